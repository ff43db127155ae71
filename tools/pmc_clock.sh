#!/bin/bash
# Effective shader clock per kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration, from one rocprofv3
# pass (counters + kernel trace) over tools/gemm_layers.py.  Usage (GPU box, repo root):  bash tools/pmc_clock.sh [ENV=..]
export TMPDIR=/tmp
root=$(pwd)
cd /tmp
rm -rf /tmp/pc0
env "$@" timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d /tmp/pc0 -o c -- python $root/tools/gemm_layers.py 8 > /dev/null 2>/tmp/pc0.err || { echo failed; tail -3 /tmp/pc0.err; }
python - <<'PY'
import csv, glob, collections
cc = glob.glob("/tmp/pc0/**/c_counter_collection.csv", recursive=True)[0]
kt = glob.glob("/tmp/pc0/**/c_kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(cc)):
    d = r["Dispatch_Id"]
    if d not in dur or "gemm" not in dur[d][1]:
        continue
    key = (dur[d][1][:52], r.get("Grid_Size", ""))
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    acc[key]["_ns_" + r["Counter_Name"]] += dur[d][0]
for key, v in sorted(acc.items()):
    ns = v["_ns_GRBM_GUI_ACTIVE"]
    clk = v["GRBM_GUI_ACTIVE"] / 8.0 / ns if ns else 0
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0
    print("%-54s grid %-8s clock %.2f GHz  MFMA busy %.0f %% of SIMD-cycles  CU busy %.0f %%" % (
        key[0], key[1], clk, 100 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc) if cyc else 0,
        100 * v["SQ_BUSY_CU_CYCLES"] / (256 * cyc) * 4 if cyc else 0))
PY
