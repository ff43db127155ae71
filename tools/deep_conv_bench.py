"""Deep VGG layers (conv3_x, conv4_x) through nfs_conv3x3_fwd / _dgrad the way the step calls them (mask of x_in from the
bit cache where the layer keeps one, an addend): kernel-only times need rocprofv3; this prints call times."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neural_flow_style_amd.ops as ops
B = 8
def timed(fn, reps=30):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
tot = [0.0, 0.0]
for HW, Ci, Co in [(50, 128, 256), (50, 256, 256), (25, 256, 512), (25, 512, 512)]:
    x = torch.relu(torch.randn(B, HW, HW, Ci, device="cuda")); w = torch.randn(3, 3, Ci, Co, device="cuda") * 0.03
    b = torch.zeros(Co, device="cuda"); wf, wd = ops.conv3x3_pack(w, 0), ops.conv3x3_pack(w, 1)
    bits = ops.conv3x3_relu_bits(B, HW, HW, Ci, Co, False, x.device)
    out = torch.empty(B, HW, HW, Co, device="cuda"); gy = torch.randn(B, HW, HW, Co, device="cuda")
    add = torch.randn(B, HW, HW, Ci, device="cuda"); gx = torch.empty_like(x)
    tf = timed(lambda: ops.conv3x3_fwd(x, wf, b, Co, True, out=out, relu_bits=bits))
    tb = timed(lambda: ops.conv3x3_dgrad(gy, wd, Ci, x_in=x, addend=add, out=gx, relu_bits=bits))
    tot[0] += tf; tot[1] += tb
    print("%2dx%-2d %3d->%-3d  fwd %6.1f us  dgrad (mask + addend) %6.1f us   bit cache: %s" % (HW, HW, Ci, Co, tf, tb, bits is not None))
print("totals fwd %.1f dgrad %.1f us" % tuple(tot))
