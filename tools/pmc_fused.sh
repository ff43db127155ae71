#!/bin/bash
# SQ counters of winograd_fused_kernel over tools/fused_conv_bench.py (GPU box, repo root): bash tools/pmc_fused.sh
export TMPDIR=/tmp
root=$(pwd)
cd /tmp
pass() {  # tag counters...
  tag=$1; shift
  rm -rf /tmp/pf_$tag
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pf_$tag -o c -- python $root/tools/fused_conv_bench.py --reps 3 > /dev/null 2>/tmp/pf_$tag.err || { echo "pass $tag failed"; tail -3 /tmp/pf_$tag.err; }
}
pass a GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES
pass b SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
pass c SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD
pass d SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for tag in "abcd":
    cc = glob.glob("/tmp/pf_%s/**/c_counter_collection.csv" % tag, recursive=True)
    kt = glob.glob("/tmp/pf_%s/**/c_kernel_trace.csv" % tag, recursive=True)
    if not cc or not kt:
        continue
    dur = {}
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    seen = set()
    for r in csv.DictReader(open(cc[0])):
        d = r["Dispatch_Id"]
        if d not in dur or "fused_kernel" not in dur[d][1]:
            continue
        key = dur[d][1].split("<")[1].split(">")[0] if "<" in dur[d][1] else dur[d][1][:40]
        key = key + " grid " + r.get("Grid_Size", "")
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if (tag, d) not in seen:
            seen.add((tag, d)); acc[key]["_ns_" + tag] += dur[d][0]; acc[key]["_n_" + tag] += 1
for key, v in sorted(acc.items()):
    print(key)
    for tag in "abcd":
        nd = v.get("_n_" + tag, 0)
        if nd:
            print("   pass %s: %d dispatches, %.1f us each" % (tag, nd, v["_ns_" + tag] / nd / 1e3))
    for name, val in sorted(v.items()):
        if not name.startswith("_"):
            tag = [t for t in "abcd" if v.get("_n_" + t)][0]
            print("   %-28s %14.0f per dispatch" % (name, val / max(1, max(v.get("_n_" + t, 0) for t in "abcd"))))
PY
