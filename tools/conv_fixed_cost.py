"""time(K) = a + b*K at fixed M x N: separates the per-launch fixed cost from the steady-state MFMA rate"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neural_flow_style_amd.ops as ops
for (B, HW, Co) in [(8, 200, 64), (8, 100, 128), (8, 50, 256), (8, 25, 512), (1, 100, 128), (1, 25, 512)]:
    pts = []
    for Ci in (32, 64, 128, 256, 512):
        x = torch.randn(B, HW, HW, Ci, device="cuda"); w = torch.randn(3, 3, Ci, Co, device="cuda") * 0.05
        b = torch.zeros(Co, device="cuda"); wf = ops.conv3x3_pack(w, 0) if Ci % 64 == 0 else None
        if wf is None:  # pack requires Ci % 64: emulate Ci=32 by packing 64 and slicing K (same kernel path)
            continue
        out = torch.empty(B, HW, HW, Co, device="cuda")
        for splitk in (False,):
            ops.conv3x3_fwd(x, wf, b, Co, True, out=out, splitk=splitk); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.conv3x3_fwd(x, wf, b, Co, True, out=out, splitk=splitk)
            e1.record(); torch.cuda.synchronize()
            pts.append((Ci, e0.elapsed_time(e1) / 10))
    (c0, t0), (c1, t1) = pts[0], pts[-1]
    bslope = (t1 - t0) / (c1 - c0); a = t0 - bslope * c0
    fl = 2.0 * B * HW * HW * 9 * Co
    print("B=%d %dx%d ->%d : %s | fixed %.1f us, marginal %.1f TF/s" % (B, HW, HW, Co, ["%d:%.0fus" % (c, t * 1e3) for c, t in pts], a * 1e3, fl / bslope / 1e9))
