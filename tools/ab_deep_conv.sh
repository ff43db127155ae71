# same-box A/B of two library builds on the deep-layer convolution calls (see tools/ab_libs.sh for the set-up)
cd neural-flow-style_amd; cp libnfs_hip.so /tmp/new.so
for i in 1 2; do
  cp libnfs_old.so libnfs_hip.so; (cd ..; echo old; python tools/deep_conv_bench.py 2>&1 | tail -5)
  cp /tmp/new.so libnfs_hip.so; (cd ..; echo HEAD; python tools/deep_conv_bench.py 2>&1 | tail -5)
done
