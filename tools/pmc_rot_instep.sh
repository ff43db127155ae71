#!/bin/bash
# SQ counters of the rotate adjoint AS THE STEP RUNS IT (the COEF instance, 8 views of 200^3) over a short bench.py run
# (separate passes, kernel-trace only, each under timeout):  bash tools/pmc_rot_instep.sh
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
BENCH="python $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-parity --no-sustained --no-other-configs --no-split-limb"
i=0
for s in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM"; do
  rm -rf /tmp/pri$i
  timeout 200 rocprofv3 --pmc $s --kernel-trace --output-format csv -d /tmp/pri$i -o c -- $BENCH > /dev/null 2>&1 || echo "pass $i failed: $s"
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("/tmp/pri*")):
    for f in glob.glob(d + "/**/c_counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "rotate_bwd_tiled_kernel<true>" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            v = sorted(v); print("%-28s n=%3d median %.5g" % (k, len(v), v[len(v)//2]))
PY
