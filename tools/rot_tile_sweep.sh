#!/bin/bash
# Sweep the rotate adjoint's tile shape / row-group width on the GPU box (compiles warp.hip per variant there).
# usage: tools/rot_tile_sweep.sh "14 12 40 8" "13 13 34 8" ...
cd "$(dirname "$0")/../neural-flow-style_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function"
cp ../libnfs_hip.so /tmp/libnfs_keep.so
OTHERS=$(ls *.o | grep -v '^warp.o$')
for cfg in "$@"; do
  set -- $cfg
  /opt/rocm/bin/hipcc $FLAGS -DNFS_RT_TZ=$1 -DNFS_RT_TY=$2 -DNFS_RT_TX=$3 -DNFS_RT_GROUP=$4 -c warp.hip -o /tmp/warp_v.o 2>/dev/null || { echo "$cfg: build failed"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS /tmp/warp_v.o -o ../libnfs_hip.so
  echo -n "tile $1 x $2 x $3 group $4: "
  python ../../tools/rot_bench.py 2>&1 | grep "rotate_bwd ms"
done
cp /tmp/libnfs_keep.so ../libnfs_hip.so
