export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
i=0
for s in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  rm -rf /tmp/pt$i
  NFS_GEMM_TUNE=0 timeout 150 rocprofv3 --pmc $s --kernel-trace --output-format csv -d /tmp/pt$i -o c -- python $root/tools/split_gemm_one.py "$@" > /dev/null 2>/tmp/pt$i.err || { echo "pass $i failed: $s"; tail -3 /tmp/pt$i.err; }
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("/tmp/pt[0-9]")):
    for f in glob.glob(d + "/**/c_counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "winograd_gemm_rb16" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            for c, vals in v.items():
                vals = sorted(vals)
                print("%-50s %-26s n=%3d median %.4g" % (k[10:], c, len(vals), vals[len(vals) // 2]))
PY
