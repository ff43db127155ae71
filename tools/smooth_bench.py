"""smooth3d + relu forward / adjoint timing at 200^3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import neural_flow_style_amd.ops as ops
from neural_flow_style_amd import synthetic as S
G = 200
d = torch.tensor(S.blob_density(G, np.random.RandomState(0)), device="cuda") - 0.01
g = torch.randn(G, G, G, device="cuda")
out = torch.empty_like(d); gd = torch.empty_like(d)
for name, f in (("fwd", lambda: ops.smooth3d_relu_fwd(d, 3.0, out=out)), ("bwd", lambda: ops.smooth3d_relu_bwd(out, g, 3.0, g_d=gd))):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("smooth3d %s %.4f ms  %.0f GB/s algorithmic (8 G^3 bytes)" % (name, ms, 8.0 * G**3 / ms / 1e6))
