"""histogram loss + gradient (hist.hip) at the shapes its callers use: the 3-channel loss-net input and conv feature maps"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neural_flow_style_amd.ops as ops

def t(f, reps=20):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for B, h, w, C in ((1, 300, 450, 3), (8, 200, 200, 3), (1, 150, 225, 64), (1, 75, 112, 256), (8, 50, 50, 256), (1, 512, 1024, 3)):
    F = torch.rand(B, h, w, C, device="cuda") * 255
    T = torch.rand(1, h, w, C, device="cuda") * 255
    l = torch.zeros(B, device="cuda"); g = torch.zeros_like(F)
    ms = t(lambda: ops.hist_loss(F, T, 1.0, l, g))
    print("hist loss + gradient  [%d, %d, %d, %d]: %.3f ms  (%.1f MB of features: %.0f GB/s for 4 passes)" % (B, h, w, C, ms, F.numel() * 4e-6, 4 * F.numel() * 4 / ms / 1e6))
