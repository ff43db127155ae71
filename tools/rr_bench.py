"""Fused rotate+render forward / render adjoint timing at the benchmark shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import neural_flow_style_amd.ops as ops
from neural_flow_style_amd import synthetic as S, transform as T
G = 200
for V in (8, 1):
    d = torch.tensor(S.blob_density(G, np.random.RandomState(0)), device="cuda")
    rot = T.rot_to_device(S.uniform_views(V), "cuda")
    d_rot = torch.empty(V, G, G, G, device="cuda")
    def f(): return ops.rotate_render_fwd(d, rot, 0.01, False, d_rot=d_rot)
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("V=%d rotate_render_fwd %.3f ms  %.0f GB/s algorithmic (8 V G^3 + 4 V G^2 bytes)" % (V, ms, (8.0 * V * G**3 + 4.0 * V * G * G) / ms / 1e6))
