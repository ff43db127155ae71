# Same-box A/B of two builds of the library: put the other build at neural-flow-style_amd/libnfs_old.so (e.g. `git worktree add /tmp/t <commit>; make -C /tmp/t/neural-flow-style_amd/csrc`), then on the GPU box: bash tools/ab_libs.sh
# (boxes differ by ~2 % in what they sustain; only runs on one box compare)
cd neural-flow-style_amd; cp libnfs_hip.so /tmp/new.so
F="--steps 40 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-parity --no-sustained --no-other-configs --no-split-limb"
for i in 1 2 3; do
  cp libnfs_old.so libnfs_hip.so; (cd ..; python bench.py $F 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('old   lib', round(d['value'],1))")
  cp /tmp/new.so libnfs_hip.so; (cd ..; python bench.py $F 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('HEAD  lib', round(d['value'],1))")
done
