"""Time of the 16-row register-B GEMM against the number of tiles of a launch (batch sweep at a fixed problem per batch
entry): the staircase tells the per-round fixed cost from the MFMA work.  Uses nfs_gram_bwd (symmetric B read in place).
    NFS_GEMM_RB=3 NFS_GEMM_BM=80 NFS_GEMM_BN=64 python tools/gemm_staircase.py [T K]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neural_flow_style_amd.ops as ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 800
K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
bm = int(os.environ.get("NFS_GEMM_BM", "80")); bn = int(os.environ.get("NFS_GEMM_BN", "64"))
for Z in (4, 8, 16, 24, 32, 40, 49, 56, 64, 80, 98, 128, 196):
    F = torch.randn(Z, T, K, device="cuda")
    D = torch.randn(Z, K, K, device="cuda"); D = D + D.transpose(1, 2)
    out = torch.empty_like(F)
    for _ in range(3): ops.gram_bwd(F, D, 1.0, relu_mask=False, out=out)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.gram_bwd(F, D, 1.0, relu_mask=False, out=out)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    tiles = Z * ((T + bm - 1) // bm) * (K // bn)
    print("Z=%3d tiles=%5d (%.2f per CU)  %7.1f us  %6.1f TF/s" % (Z, tiles, tiles / 256.0, best * 1e3, 2.0 * Z * T * K * K / best / 1e9))
