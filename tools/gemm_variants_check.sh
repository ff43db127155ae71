# the conv tests over every GEMM kernel x tile combination (GPU box, repo root): bash tools/gemm_variants_check.sh
for rb in 0 2; do for bm in 64 128; do for bn in 64 128; do
  r=$(NFS_WG_FUSED=0 NFS_GEMM_RB=$rb NFS_GEMM_BM=$bm NFS_GEMM_BN=$bn timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "conv3x3_fwd_and_dgrad or fused_pool" 2>&1 | tail -1)
  echo "rb=$rb bm=$bm bn=$bn: $r"
done; done; done
for bm in 80 48 112 208; do for bn in 64 128; do
  r=$(NFS_WG_FUSED=0 NFS_GEMM_RB=3 NFS_GEMM_BM=$bm NFS_GEMM_BN=$bn timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "conv3x3_fwd_and_dgrad or fused_pool" 2>&1 | tail -1)
  echo "rb16 bm=$bm bn=$bn: $r"
done; done
