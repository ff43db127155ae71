#!/bin/bash
# Evidence run for profiles/: kernel-trace stats, the two PMC passes (separate runs, kernel-trace only), and the
# plain bench line.  Usage (on the GPU box, from the repo root):  bash tools/profile_round.sh r01_c
# Writes gpurun_out/<tag>_*; copy what should be judged into profiles/.
set -u
tag=${1:-r01_x}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
BENCH="python $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-parity --no-sustained --no-other-configs --no-split-limb --no-skip-control"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o b -- $BENCH > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_pmc_fetch -o b -- $BENCH > /dev/null 2> $out/${tag}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_pmc_write -o b -- $BENCH > /dev/null 2> $out/${tag}_pmc_write.err
cd $root
f=$(find $out/${tag}_stats -name 'b_kernel_stats.csv' | head -1)
cp "$f" $out/${tag}_bench_kernel_stats.csv
# in-step averages over the 10 TIMED steps (set-up launches, the warm-up steps and the tile tuner's trial launches in
# them excluded): what roofline.avg_launch_us has to agree with
python tools/instep_stats.py $(find $out/${tag}_stats -name 'b_kernel_trace.csv' | head -1) 10 $out/${tag}_instep.json > $out/${tag}_instep.txt
python tools/pmc_traffic.py $(find $out/${tag}_pmc_fetch -name 'b_counter_collection.csv' | head -1) \
                            $(find $out/${tag}_pmc_write -name 'b_counter_collection.csv' | head -1) \
                            $out/${tag}_traffic.json > $out/${tag}_traffic.txt
# the raw counter CSVs are large; keep only the summaries
rm -rf $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_stats
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -c 600 $out/${tag}_bench_under_rocprof.json; cat $out/${tag}_instep.txt
python - <<EOF
import csv
rows=list(csv.DictReader(open("$out/${tag}_bench_kernel_stats.csv")))
for r in rows[:16]:
    print("%-60s calls %6s avg_us %9.1f pct %5s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
EOF
