"""The render family (rotate+render forward, render adjoint, rotate adjoint) at 200^3 for 1, 2, 4, 8 views, each kernel
alone and back to back: microseconds per launch and per view -- how much of the one-view chain is the family's own
small-launch inefficiency."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import neural_flow_style_amd.ops as ops
from neural_flow_style_amd import synthetic as S, transform as T

G = 200
d = torch.tensor(S.blob_density(G, np.random.RandomState(0)), device="cuda")
mats = S.uniform_views(8)


def timeit(f, reps=20):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3


for V in (1, 2, 4, 8):
    rot = T.rot_to_device(mats[:V], "cuda")
    g = torch.randn(V, G, G, G, 1, device="cuda", generator=torch.Generator(device="cuda").manual_seed(11))
    acc = torch.zeros(G, G, G, 1, device="cuda")
    gmax = g.abs().max().reshape(1)
    t_rb = timeit(lambda: ops.rotate_bwd(g, rot, g_d_acc=acc, g_max=gmax))
    d_rot = torch.empty(V, G, G, G, device="cuda")
    img = torch.empty(V, G, G, device="cuda"); rs = torch.empty(V, G, G, device="cuda")
    t_rr = timeit(lambda: ops.rotate_render_fwd(d, rot, 0.01, 0, img=img, raysum=rs, d_rot=d_rot))
    g_img = torch.randn(V, G, G, device="cuda")
    keep = d_rot.clone()
    gd = torch.empty_like(d_rot)
    t_re = timeit(lambda: ops.render_bwd(keep, rs, g_img, 0.01, 0, g_d=gd, want_max=True))
    print("V=%d rotate_bwd %7.1f us (%5.1f /view)  rotate_render_fwd %6.1f us (%5.1f /view)  render_bwd %6.1f us (%5.1f /view)"
          % (V, t_rb, t_rb / V, t_rr, t_rr / V, t_re, t_re / V))
    if ops.render_coef_layout(V, G, G, G) is not None:          # the u / coefficient form of the adjoint
        _, _, u_rot, seg = ops.rotate_render_fwd_coef(d, rot, 0.01, img=img, raysum=rs, u_rot=d_rot)
        t_fc = timeit(lambda: ops.rotate_render_fwd_coef(d, rot, 0.01, img=img, raysum=rs, u_rot=d_rot, seg=seg))
        ab, gm = ops.render_ray_coef(g_img, seg, 0.01)
        t_co = timeit(lambda: ops.render_ray_coef(g_img, seg, 0.01, ab=ab, bounds=gm))
        t_bc = timeit(lambda: ops.rotate_bwd_coef(u_rot, ab, rot, gm, g_d_acc=acc[..., 0]))
        print("     coefficient form: forward %6.1f us  ray coefficients %5.1f us  rotate adjoint %6.1f us" % (t_fc, t_co, t_bc))
