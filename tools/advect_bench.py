"""advect forward / velocity-gradient timing at 200^3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import neural_flow_style_amd.ops as ops
from neural_flow_style_amd import synthetic as S
G = 200
rng = np.random.RandomState(0)
d = torch.tensor(S.blob_density(G, rng), device="cuda")[..., None].contiguous()
vel = torch.randn(G, G, G, 3, device="cuda") * (0.02 / (G - 1))   # early-iteration velocities: a small fraction of a cell
g = torch.randn(G, G, G, 1, device="cuda")
out = torch.empty_like(d); gv = torch.empty_like(vel)
for name, f, nbytes in (("fwd", lambda: ops.advect_fwd(d, vel, out=out), 20.0 * G**3),
                        ("bwd(vel)", lambda: ops.advect_bwd(d, vel, g, need_d=False, g_vel=gv), 32.0 * G**3)):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("advect %s %.4f ms  %.0f GB/s algorithmic" % (name, ms, nbytes / ms / 1e6))
m = torch.zeros_like(vel); v2 = torch.zeros_like(vel)
def fa(): ops.advect_bwd_adam(d, vel, g, m, v2, 1e-9, 0.9, 0.999, 1e-8)
fa(); torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fa()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
print("advect bwd+adam %.4f ms  %.0f GB/s algorithmic (80 B/voxel)" % (best, 80.0 * G**3 / best / 1e6))
