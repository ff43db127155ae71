"""Narrow layers (single-kernel Winograd): one launch over 8 views against two concurrent half-batches on two streams
(the tail of one half's kernel is filled by the other half's blocks)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neural_flow_style_amd.ops as ops

B = 8
def mk(HW, Ci, Co):
    x = torch.relu(torch.randn(B, HW, HW, Ci, device="cuda")); w = torch.randn(3, 3, Ci, Co, device="cuda") * 0.05
    return x, ops.conv3x3_pack(w, 0), torch.zeros(Co, device="cuda"), torch.empty(B, HW, HW, Co, device="cuda"), Co
layers = [mk(200, 64, 64), mk(100, 64, 128), mk(100, 128, 128)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def whole():
    for x, wf, b, out, Co in layers:
        ops.conv3x3_fwd(x, wf, b, Co, True, out=out)
def halves(nsplit=2):
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(cur)
    h = B // nsplit
    for i, s in enumerate((s1, s2)[:nsplit]):
        s.wait_event(ev)
        with torch.cuda.stream(s):
            for x, wf, b, out, Co in layers:
                ops.conv3x3_fwd(x[i * h:(i + 1) * h], wf, b, Co, True, out=out[i * h:(i + 1) * h])
    for s in (s1, s2)[:nsplit]:
        e = torch.cuda.Event(); e.record(s); cur.wait_event(e)
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("conv1_2 + conv2_1 + conv2_2 forward, 8 views: one stream %.1f us, two half-batches on two streams %.1f us" % (timed(whole), timed(halves)))
for li, name in enumerate(("conv1_2", "conv2_1", "conv2_2")):
    keep = layers
    layers = [keep[li]]
    print("  %s alone: %.1f us / %.1f us" % (name, timed(whole), timed(halves)))
    layers = keep
