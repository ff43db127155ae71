"""Batched f32-MFMA GEMM (winograd_gemm_kernel) through nfs_gram_bwd: C[z] = F[z] (T x K) @ D[z] (K x K)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neural_flow_style_amd.ops as ops

for Z, T, K in [(36, 1352, 256), (36, 1408, 256), (36, 1408, 512), (36, 1408, 1024), (8, 5632, 2048), (36, 392, 512),
                (36, 5000, 128), (144, 1408, 256), (16, 5000, 256)]:
    F = torch.randn(Z, T, K, device="cuda")
    D = torch.randn(Z, K, K, device="cuda")
    out = torch.empty_like(F)
    ops.gram_bwd(F, D, 1.0, relu_mask=False, out=out); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gram_bwd(F, D, 1.0, relu_mask=False, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("Z=%3d T=%5d K=N=%4d  %7.3f ms  %6.1f TF/s" % (Z, T, K, ms, 2.0 * Z * T * K * K / ms / 1e9))
