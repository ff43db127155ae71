#!/bin/bash
# TIMING-ONLY: the narrow-layer kernel with its slice loop on the bf16 pipe's cycle count (NFS_K7F_EMUL_BF16: 108 bf16 MFMAs
# per slice instead of 144 f32 ones + the limb split in the transform; results are wrong by construction) against the
# shipped kernel, same box.  What the verdict's split-limb K7f could gain at best BEFORE its filter stream and its
# one-block-per-CU block shape are paid for.   usage (GPU box, repo root): bash tools/k7f_bf16_emul.sh
cd "$(dirname "$0")/../neural-flow-style_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function"
echo "shipped kernel:"; python ../../tools/fused_conv_bench.py 2>&1 | grep -v amdgpu
cp ../libnfs_hip.so /tmp/libnfs_keep.so
OTHERS=$(ls *.o | grep -v '^winograd_fused.o$')
/opt/rocm/bin/hipcc $FLAGS -DNFS_K7F_EMUL_BF16 -c winograd_fused.hip -o /tmp/wf_emul.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS /tmp/wf_emul.o -o ../libnfs_hip.so
echo "bf16-pipe emulation (wrong results):"; python ../../tools/fused_conv_bench.py 2>&1 | grep -v amdgpu
cp /tmp/libnfs_keep.so ../libnfs_hip.so
