set -u
root=$(pwd); out=$root/gpurun_out; export TMPDIR=/tmp
cd /tmp
for g in 0 1; do
ONE_VIEW_GRAPH=$g rocprofv3 --kernel-trace --stats --output-format csv -d $out/ov$g -o b -- python $root/tools/one_view_profile.py 1 100 > /dev/null 2> $out/ov$g.err
done
cd $root
for g in 0 1; do
f=$(find $out/ov$g -name 'b_kernel_trace.csv' | head -1)
python tools/timeline.py $f 3 > $out/ov${g}_timeline.txt
python tools/gap_stats.py $f 50 > $out/ov${g}_gaps.txt
cp $(find $out/ov$g -name 'b_kernel_stats.csv' | head -1) $out/ov${g}_kernel_stats.csv
rm -rf $out/ov$g
done
tail -3 $out/ov1_timeline.txt; head -3 $out/ov1_gaps.txt; head -3 $out/ov0_gaps.txt
