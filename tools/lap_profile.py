"""Laplacian-pyramid normalisation of a 200^3 x 3 gradient field: event timing of its pieces per pyramid level."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_flow_style_amd import util as U, ops

def t(f, reps=10):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

g = torch.randn(200, 200, 200, 3, device="cuda")
k = torch.as_tensor(U.lap_kernel(True)).cuda().contiguous()
cur = g
for lvl in range(3):
    lo = ops.lap_down(cur, k)
    print("level %d (%s -> %s): lap_down %.3f ms, lap_up + addend %.3f ms, normalize_mean %.3f ms" % (
        lvl, tuple(cur.shape[:3]), tuple(lo.shape[:3]), t(lambda: ops.lap_down(cur, k)),
        t(lambda: ops.lap_up(lo, k, cur.shape, -5.0, addend=cur)), t(lambda: ops.normalize_mean(cur, use_abs=False, eps=1e-10))))
    cur = lo
print("whole lap_normalize(scale_n=3): %.3f ms" % t(lambda: U.lap_normalize(g, scale_n=3, is_3d=True, c=3), 5))
# reference points for level 0: a plain 96 MB -> 96 MB elementwise pass, and the transpose without its addend
lo0 = ops.lap_down(g, k)
print("torch.add 96 MB -> 96 MB: %.3f ms; lap_up without addend: %.3f ms; with: %.3f ms" % (
    t(lambda: torch.add(g, 1.0)), t(lambda: ops.lap_up(lo0, k, g.shape, -5.0)), t(lambda: ops.lap_up(lo0, k, g.shape, -5.0, addend=g))))
