"""Per-layer time of the Gram chain (gram_fwd = tn kernel + slab reduce, style loss, gram_bwd) at the headline shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neural_flow_style_amd.ops as ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
tot = [0, 0, 0]
for HW, C in [(200, 64), (100, 128), (50, 256), (25, 512), (12, 512)]:
    F = torch.relu(torch.randn(B, HW, HW, C, device="cuda"))
    Gs = torch.randn(1, C, C, device="cuda")
    loss = torch.zeros(B, device="cuda")
    G = ops.gram_fwd(F, 1.0 / (HW * HW))
    D = ops.style_loss_fwd(G, Gs, 1.0, loss)
    out = torch.empty_like(F)
    t_f = timed(lambda: ops.gram_fwd(F, 1.0 / (HW * HW), G=G))
    t_l = timed(lambda: ops.style_loss_fwd(G, Gs, 1.0, loss, Dmat=D))
    t_b = timed(lambda: ops.gram_bwd(F, D, 1.0, out=out))
    fl = 2.0 * B * HW * HW * C * C
    by = 4.0 * B * HW * HW * C
    print("%3dx%-3d C=%3d  gram_fwd %6.1f us (%5.1f TF/s, F read %5.2f TB/s)  style_loss %5.1f us  gram_bwd %6.1f us (%5.1f TF/s, %5.2f TB/s)" % (
        HW, HW, C, t_f, fl / t_f / 1e6, by / t_f / 1e6, t_l, t_b, fl / t_b / 1e6, 2 * by / t_b / 1e6))
    tot[0] += t_f; tot[1] += t_l; tot[2] += t_b
print("totals: gram_fwd %.1f  style_loss %.1f  gram_bwd %.1f us" % tuple(tot))

# the grouped form: all five layers in three launches (tile pairs, slab reduce + style loss, one batched Gram-gradient GEMM)
import ctypes
from neural_flow_style_amd import _lib
Fs, Gss = [], []
for HW, C in [(200, 64), (100, 128), (50, 256), (25, 512), (12, 512)]:
    Fs.append(torch.relu(torch.randn(B, HW, HW, C, device="cuda")))
    Gss.append(torch.randn(1, C, C, device="cuda"))
t_all = timed(lambda: ops.gram_style_group(Fs, Gss, [1.0] * 5, [True] * 5))
L = _lib.lib()
import neural_flow_style_amd.ops as O2
_lib.PROFILE = {}
for _ in range(20): ops.gram_style_group(Fs, Gss, [1.0] * 5, [True] * 5)
torch.cuda.synchronize()
for k, v in _lib.PROFILE.items():
    print("  %-28s %7.1f us (event pair included)" % (k, sum(a.elapsed_time(b) for a, b, _ in v) / len(v) * 1e3))
_lib.PROFILE = None
print("grouped: fwd + loss + bwd of the five layers %.1f us per call (was %.1f)" % (t_all, sum(tot)))
