#!/bin/bash
export TMPDIR=/tmp
cd /tmp
# SQ counters of the batched GEMM over tools/gemm_layers.py (separate passes, kernel-trace only; each pass under timeout)
sets=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
      "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"
      "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU")
i=0
for s in "${sets[@]}"; do
  rm -rf /tmp/pg$i
  timeout 150 rocprofv3 --pmc $s --kernel-trace --output-format csv -d /tmp/pg$i -o c -- python /root/repo/tools/gemm_layers.py 8 > /dev/null 2>/tmp/pg$i.err || { echo "pass $i failed: $s"; tail -3 /tmp/pg$i.err; }
  i=$((i+1))
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("/tmp/pg[0-9]")):
    for f in glob.glob(d + "/**/c_counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "winograd_gemm_kernel" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(k)
            for c, vals in v.items():
                print("   %-34s n=%3d sum %.4g" % (c, len(vals), sum(vals)))
PY
