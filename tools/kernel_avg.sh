#!/bin/bash
# average kernel durations (rocprofv3 --kernel-trace --stats) of the kernels whose names match a pattern, for one command:
#   tools/kernel_avg.sh 'advect1|winograd_input4' python tools/one_view_time.py 8
pat=$1; shift
export TMPDIR=/tmp
d=$(mktemp -d /tmp/kavg.XXXXXX)
( cd /tmp && ONE_VIEW_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o k -- "$@" > /dev/null 2>&1 )
python3 - "$pat" $(find $d -name 'k_kernel_stats.csv' | head -1) <<'PY'
import csv, re, sys
pat = re.compile(sys.argv[1])
for r in csv.DictReader(open(sys.argv[2])):
    if pat.search(r["Name"]):
        print("  %-72s calls %6s avg %8.2f us" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $d
