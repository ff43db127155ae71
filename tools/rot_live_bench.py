"""The rotate adjoint of the benchmark step alone, coefficient form, with and without the live mask of the benchmark's own
density / velocity (200^3, 8 views): ms per call (event pairs over 20 calls, best of 4), incl. the two box launches.
    python tools/rot_live_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import neural_flow_style_amd.ops as ops
from neural_flow_style_amd import synthetic as S, transform as T
G, V = 200, 8
rng = np.random.RandomState(123)
d0 = torch.tensor(S.blob_density(G, rng), device="cuda")
vel = torch.tensor(S.curl_velocity(G, rng, max_cells=2.0), device="cuda")
live = ops.live_mask(G, G, G, d0)
d_adv = ops.advect_fwd(d0.unsqueeze(-1), vel, live=live).squeeze(-1)
d_s = ops.smooth3d_relu_fwd(d_adv, 3.0)
rot = T.rot_to_device(S.uniform_views(V), "cuda")
u_rot = torch.empty((V, G, G, G), dtype=torch.float32, device="cuda")
img, rs, _, seg = ops.rotate_render_fwd_coef(d_s, rot, 0.01, u_rot=u_rot)
g_img = torch.randn(V, G, G, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
ab, bounds = ops.render_ray_coef(g_img, seg, 0.01)
out = torch.empty(G, G, G, device="cuda")
def t(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    return best
full = t(lambda: ops.rotate_bwd_coef(u_rot, ab, rot, bounds, g_d_acc=out, overwrite=True))
ref = out.clone()
masked = t(lambda: ops.rotate_bwd_coef(u_rot, ab, rot, bounds, g_d_acc=out, overwrite=True, live=live, dilate=1))
print("rotate adjoint: whole volume %.4f ms, live boxes %.4f ms" % (full, masked))
