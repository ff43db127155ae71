export TMPDIR=/tmp
root=$(pwd)
cd /tmp
i=0
for s in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU"; do
  rm -rf /tmp/pr$i
  timeout 150 rocprofv3 --pmc $s --kernel-trace --output-format csv -d /tmp/pr$i -o c -- python $root/tools/rot_bench.py > /dev/null 2>&1 || echo "pass $i failed: $s"
  i=$((i+1))
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("/tmp/pr*")):
    for f in glob.glob(d + "/**/c_counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "rotate_bwd_tiled" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            v = sorted(v); print("%-28s n=%3d median %.5g" % (k, len(v), v[len(v)//2]))
PY
