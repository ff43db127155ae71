#!/bin/bash
# SQ counters of the split-limb 16-row GEMM (rb16s) over tools/split_gemm_one.py (separate passes, kernel-trace only;
# each pass under timeout):  bash tools/pmc_rb16s.sh [args of split_gemm_one.py]
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
sets=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
      "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"
      "SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM")
i=0
for s in "${sets[@]}"; do
  rm -rf /tmp/ps$i
  NFS_GEMM_TUNE_LOG=1 timeout 150 rocprofv3 --pmc $s --kernel-trace --output-format csv -d /tmp/ps$i -o c -- python $root/tools/split_gemm_one.py "$@" > /dev/null 2>/tmp/ps$i.err || { echo "pass $i failed: $s"; tail -3 /tmp/ps$i.err; }
  grep "gemm tuner" /tmp/ps$i.err | head -2
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("/tmp/ps[0-9]")):
    for f in glob.glob(d + "/**/c_counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "winograd_gemm_rb16" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(k)
            for c, vals in v.items():
                vals = sorted(vals)
                print("   %-34s n=%3d median %.4g" % (c, len(vals), vals[len(vals) // 2]))
PY
