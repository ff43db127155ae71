import sys, os
sys.path.insert(0, "/root/repo")
import torch
import neural_flow_style_amd.ops as ops
def timed(fn, reps=30):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for B, HW in [(1, 200), (1, 100), (2, 100), (1, 48), (8, 48), (3, 128)]:
    for Ci, Co in [(64, 64), (64, 128), (128, 128)]:
        H = HW if Ci == 64 and Co == 64 else HW // 2
        x = torch.relu(torch.randn(B, H, H, Ci, device="cuda")); w = torch.randn(3, 3, Ci, Co, device="cuda") * 0.05
        b = torch.zeros(Co, device="cuda"); wf = ops.conv3x3_pack(w, 0); wd = ops.conv3x3_pack(w, 1)
        out = torch.empty(B, H, H, Co, device="cuda"); gy = torch.randn(B, H, H, Co, device="cuda")
        tf = timed(lambda: ops.conv3x3_fwd(x, wf, b, Co, True, out=out))
        tb = timed(lambda: ops.conv3x3_dgrad(gy, wd, Ci))
        print("B=%d %3dx%-3d %3d->%-3d tiles %6d  fwd %6.1f us  dgrad %6.1f us" % (B, H, H, Ci, Co, B * (H // 4) ** 2, tf, tb))
