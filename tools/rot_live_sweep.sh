#!/bin/bash
# Sweep the rotate adjoint's tile shape / row-group width / block size WITH the live mask on the GPU box (compiles
# warp.hip per variant there).  usage: tools/rot_live_sweep.sh "14 14 34 4 1024" "14 14 17 4 1024" ...
cd "$(dirname "$0")/../neural-flow-style_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function"
cp ../libnfs_hip.so /tmp/libnfs_keep.so
OTHERS=$(ls *.o | grep -v '^warp.o$')
for cfg in "$@"; do
  set -- $cfg
  /opt/rocm/bin/hipcc $FLAGS -DNFS_RT_TZ=$1 -DNFS_RT_TY=$2 -DNFS_RT_TX=$3 -DNFS_RT_GROUP=$4 -DNFS_RT_THREADS=$5 -c warp.hip -o /tmp/warp_v.o 2>/dev/null || { echo "$cfg: build failed"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS /tmp/warp_v.o -o ../libnfs_hip.so
  echo -n "tile $1 x $2 x $3 group $4 threads $5: "
  python ../../tools/rot_live_bench.py 2>&1 | grep "rotate adjoint"
done
cp /tmp/libnfs_keep.so ../libnfs_hip.so
