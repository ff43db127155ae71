#!/bin/bash
# L2 / L1 traffic and MFMA busy counters of winograd_fused_kernel over tools/fused_conv_bench.py (GPU box, repo root)
export TMPDIR=/tmp
root=$(pwd)
cd /tmp
pass() {
  tag=$1; shift
  rm -rf /tmp/pl_$tag
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pl_$tag -o c -- python $root/tools/fused_conv_bench.py --reps 3 > /dev/null 2>/tmp/pl_$tag.err || { echo "pass $tag failed"; tail -3 /tmp/pl_$tag.err; }
}
pass a GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
pass b TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass c TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
pass d FETCH_SIZE
pass e TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
pass f SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD
pass g SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for tag in "abcdefg":
    cc = glob.glob("/tmp/pl_%s/**/c_counter_collection.csv" % tag, recursive=True)
    kt = glob.glob("/tmp/pl_%s/**/c_kernel_trace.csv" % tag, recursive=True)
    if not cc or not kt:
        print("pass", tag, "no output"); continue
    dur = {}
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    seen = set()
    for r in csv.DictReader(open(cc[0])):
        d = r["Dispatch_Id"]
        if d not in dur or "winograd_fused_kernel<" not in dur[d][1]:
            continue
        key = dur[d][1].split("<")[1].split(">")[0] + " grid " + r.get("Grid_Size", "")
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if (tag, d) not in seen:
            seen.add((tag, d)); acc[key]["_ns_" + tag] += dur[d][0]; acc[key]["_n_" + tag] += 1
for key, v in sorted(acc.items()):
    print(key)
    tags = [t for t in "abcdefg" if v.get("_n_" + t)]
    print("   us per dispatch by pass:", " ".join("%s=%.1f" % (t, v["_ns_" + t] / v["_n_" + t] / 1e3) for t in tags))
    nd = max(v.get("_n_" + t, 0) for t in "abcdefg")
    for name, val in sorted(v.items()):
        if not name.startswith("_"):
            print("   %-32s %16.0f per dispatch" % (name, val / max(1, nd)))
PY
