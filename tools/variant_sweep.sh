#!/bin/bash
# Compile ONE csrc/*.hip with different -D flags on the GPU box and run a bench command against each variant.
# usage: tools/variant_sweep.sh field.hip "python tools/smooth_bench.py" "-DNFS_SM_PF=2" "-DNFS_SM_PF=4" ...
src=$1; cmd=$2; shift 2
root="$(cd "$(dirname "$0")/.." && pwd)"
cd "$root/neural-flow-style_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function"
cp ../libnfs_hip.so /tmp/libnfs_keep.so
obj="${src%.hip}.o"
OTHERS=$(ls *.o | grep -v "^$obj\$")
for cfg in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS $cfg -c $src -o /tmp/variant.o 2>/dev/null || { echo "$cfg: build failed"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS /tmp/variant.o -o ../libnfs_hip.so
  echo "== $cfg"
  (cd "$root" && eval "$cmd" 2>&1 | grep -v "amdgpu.ids")
done
cp /tmp/libnfs_keep.so ../libnfs_hip.so
