"""How does the library f32 batched GEMM (torch.bmm -> rocBLAS / hipBLASLt) do on the Winograd shapes?"""
import torch
for name, Z, T, K, N in [("conv1_2", 36, 20000, 64, 64), ("conv2_1", 36, 5000, 64, 128), ("conv2_2", 36, 5000, 128, 128),
                         ("conv3_1", 36, 1352, 128, 256), ("conv3_x", 36, 1352, 256, 256), ("conv4_1", 36, 392, 256, 512),
                         ("conv4_x", 36, 392, 512, 512), ("conv5_1", 36, 72, 512, 512)]:
    A = torch.randn(Z, T, K, device="cuda"); B = torch.randn(Z, K, N, device="cuda"); C = torch.empty(Z, T, N, device="cuda")
    for _ in range(3): torch.bmm(A, B, out=C)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): torch.bmm(A, B, out=C)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%-8s Z=%d T=%5d K=%3d N=%3d  bmm %7.1f us  %6.1f TF/s" % (name, Z, T, K, N, ms * 1e3, 2.0 * Z * T * K * N / ms / 1e9))
