"""SPH splat (p2g) forward / adjoint timing at the chocolate scale (BASELINE configs[4]: ~5e5 particles)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import neural_flow_style_amd.ops as ops
from neural_flow_style_amd import synthetic as S

N, G = 500000, 200
rng = np.random.RandomState(0)
p = torch.tensor(S.blob_particles(N, rng), device="cuda")
for nsize in (1, 2):
    cfg = ops.make_splat_cfg(3, [G, G, G], [G, G, G], 0.5, 4, 1000.0, nsize, False, 0)   # mode 0: density splat
    g = torch.randn(G, G, G, 1, device="cuda")
    def t(f, reps=10):
        f(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): f()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
        return best
    cells = (2 * nsize + 1) ** 3
    tf = t(lambda: ops.p2g_fwd(p, cfg))
    tb = t(lambda: ops.p2g_bwd(p, cfg, g, need_p=True))
    print("nsize %d (%d cells/particle): p2g fwd %.3f ms (%.1f G cell-updates/s, incl. the 32 MB grid clear)  "
          "bwd %.3f ms (%.1f G cell-gathers/s)" % (nsize, cells, tf, N * cells / tf / 1e6, tb, N * cells / tb / 1e6))
# the same particles in cell order (z, y, x): neighbouring lanes then update the same cache lines
cell = (p * G).floor().clamp(0, G - 1).long()
key = (cell[:, 0] * G + cell[:, 1]) * G + cell[:, 2]
ps = p[torch.argsort(key)].contiguous()
cfg = ops.make_splat_cfg(3, [G, G, G], [G, G, G], 0.5, 4, 1000.0, 1, False, 0)
g = torch.randn(G, G, G, 1, device="cuda")
print("cell-sorted particles (row-major cells), nsize 1: fwd %.3f ms  bwd %.3f ms" % (t(lambda: ops.p2g_fwd(ps, cfg)), t(lambda: ops.p2g_bwd(ps, cfg, g, need_p=True))))
from neural_flow_style_amd import transform as T
for brick in (4, 8, 16):
    pb = p[T.grid_order(p, [G, G, G], brick)].contiguous()
    line = "grid order in %d-cell bricks, nsize 1: fwd %.3f ms  bwd %.3f ms" % (
        brick, t(lambda: ops.p2g_fwd(pb, cfg)), t(lambda: ops.p2g_bwd(pb, cfg, g, need_p=True)))
    # ... and after a Lagrangian run has moved every particle by up to +-2 / +-4 cells
    for drift in (2, 4):
        pd_ = (pb + (torch.rand_like(pb) * 2 - 1) * drift / G).clamp(0.01, 0.99)
        line += "; drifted +-%d cells: fwd %.3f" % (drift, t(lambda: ops.p2g_fwd(pd_, cfg)))
    print(line)
psd = (ps + (torch.rand_like(ps) * 2 - 1) * 2 / G).clamp(0.01, 0.99)
print("row-major cell order drifted +-2 cells: fwd %.3f ms" % t(lambda: ops.p2g_fwd(psd, cfg)))
pu = torch.rand(N, 3, device="cuda") * 0.9 + 0.05
print("uniform random particles, nsize 1: fwd %.3f ms  bwd %.3f ms" % (t(lambda: ops.p2g_fwd(pu, cfg)), t(lambda: ops.p2g_bwd(pu, cfg, g, need_p=True))))
# a later frame of a sequence: the same particles carried along by a flow (here a rotation about the axis by 0.4 rad), still
# in frame 0's order -- and seen through that frame's own grid order (what Styler.run does: gather in, scatter out)
pb = p[T.grid_order(p, [G, G, G])].contiguous()
c, s_ = float(np.cos(0.4)), float(np.sin(0.4))
q = pb - 0.5
moved = torch.stack([q[:, 0], c * q[:, 1] - s_ * q[:, 2], s_ * q[:, 1] + c * q[:, 2]], 1).mul(0.7).add(0.5).contiguous()
order = T.grid_order(moved, [G, G, G])
def via_order():
    return ops.p2g_fwd(moved[order].contiguous(), cfg)
print("frame carried by a coherent flow, frame-0 order: fwd %.3f ms; through its own grid order (incl. the gather): %.3f ms"
      % (t(lambda: ops.p2g_fwd(moved, cfg)), t(via_order)))
# ... and with mixing: every particle also wanders by up to +-10 cells relative to its frame-0 neighbours
moved = (moved + (torch.rand_like(moved) * 2 - 1) * 10 / G).clamp(0.02, 0.98).contiguous()
order = T.grid_order(moved, [G, G, G])
print("frame with mixing (+-10 cells), frame-0 order: fwd %.3f ms; through its own grid order (incl. the gather): %.3f ms"
      % (t(lambda: ops.p2g_fwd(moved, cfg)), t(via_order)))
