#!/bin/bash
# rotate adjoint: tile -> XCD assignment (NFS_RT_XCD 0 round-robin / 1 z-major chunks / 2 y-major chunks): time and HBM fetch
export TMPDIR=/tmp
root=$(pwd)
for o in 0 1 2; do
  echo "== NFS_RT_XCD=$o"
  NFS_RT_XCD=$o python tools/rot_bench.py 2>/dev/null | grep -E "ms|sha1"
done
cd /tmp
for o in 0 1 2; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/prx
    NFS_RT_XCD=$o timeout 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/prx -o c -- python $root/tools/rot_bench.py > /dev/null 2>&1 || echo "pass failed"
    python - "$o" "$c" <<'PY'
import csv, glob, sys
v = []
for f in glob.glob("/tmp/prx/**/c_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rotate_bwd_tiled" in r["Kernel_Name"]:
            v.append(float(r["Counter_Value"]))
v.sort()
print("NFS_RT_XCD=%s %s median %.1f (KB per launch, raw counter; FETCH_SIZE x2 on gfx950 per MI355X_MICROARCH.md)" % (sys.argv[1], sys.argv[2], v[len(v) // 2] if v else -1))
PY
  done
done
