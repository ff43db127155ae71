"""conv1_1 (3 -> 64 channels) forward and data gradient at the headline shape (8 x 200 x 200), HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hashlib, torch
import neural_flow_style_amd.ops as ops
B, H = (int(sys.argv[1]) if len(sys.argv) > 1 else 8), 200
x = torch.randn(B, H, H, 3, device="cuda"); w = torch.randn(3, 3, 3, 64, device="cuda") * 0.1
wf, wd = ops.conv3x3_pack(w, 0), ops.conv3x3_pack(w, 1)
y = torch.empty(B, H, H, 64, device="cuda"); gy = torch.randn(B, H, H, 64, device="cuda"); gx = torch.empty(B, H, H, 3, device="cuda")
bias = torch.zeros(64, device="cuda")
def timed(f, reps=30):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3
tf = timed(lambda: ops.conv3x3_fwd(x, wf, bias, 64, True, out=y))
tb = timed(lambda: ops.conv3x3_dgrad(gy, wd, 3, out=gx))
print("B=%d conv1_1 fwd %.1f us  dgrad %.1f us  digest %s" % (B, tf, tb, hashlib.sha1(gx.cpu().numpy().tobytes()).hexdigest()[:12]))
