#!/bin/bash
# Per-kernel PMC table for one command (several passes, kernel-trace only):
#   bash tools/pmc_kernel.sh <kernel-substring> <cmd...>
# Prints the median counter values over the launches whose name contains the substring.
# TA_* / TCP_* counter passes hang rocprofv3 on this pool (timed out twice): SQ / TCC / GRBM only.
sub=$1; shift
export TMPDIR=/tmp
cd /tmp
sets=("SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
      "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum")
i=0
for s in "${sets[@]}"; do
  rm -rf /tmp/pk$i
  timeout 150 rocprofv3 --pmc $s --kernel-trace --output-format csv -d /tmp/pk$i -o c -- "$@" > /dev/null 2>&1 || echo "pass $i failed/timed out: $s"
  i=$((i+1))
done
python - "$sub" <<'PY'
import csv, glob, sys, collections
sub = sys.argv[1]
for d in sorted(glob.glob("/tmp/pk*")):
    for f in glob.glob(d + "/**/c_counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            v = sorted(v)
            print("%-36s n=%3d median %.4g" % (k, len(v), v[len(v) // 2]))
PY
