"""Render adjoint (in place on the kept rotated volume) timing at the benchmark shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neural_flow_style_amd.ops as ops
G, V = 200, 8
d_rot = torch.rand(V, G, G, G, device="cuda") * 0.1
keep = d_rot.clone()
rs = d_rot.sum(1)
gi = torch.randn(V, G, G, device="cuda")
def f(): return ops.render_bwd(d_rot, rs, gi, 0.01, False, g_d=d_rot, want_max=True)
f(); torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    d_rot.copy_(keep)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
print("render_bwd in place %.3f ms  %.0f GB/s (8 V G^3 bytes)" % (best, 8.0 * V * G**3 / best / 1e6))
