"""smooth3d + clamp: 8-row against 16-row tiles (NFS_SM_ROWS, read once per process): time and a bit-exact digest of both
directions at a few sizes.  Run twice: NFS_SM_ROWS=8 python tools/smooth_rows_ab.py; NFS_SM_ROWS=16 ... and diff the digests."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import neural_flow_style_amd.ops as ops

def digest(t):
    return hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:12]

for shape in ((200, 200, 200), (100, 100, 100), (37, 61, 130), (16, 16, 16), (64, 200, 55)):
    g = torch.Generator(device="cuda").manual_seed(5)
    d = torch.randn(*shape, device="cuda", generator=g) * 0.5 + 0.1
    go = torch.randn(*shape, device="cuda", generator=g)
    out = ops.smooth3d_relu_fwd(d, 3.0)
    gd = ops.smooth3d_relu_bwd(out, go, 3.0)
    w1 = torch.tensor([1.0, 3.0, 1.0], device="cuda", dtype=torch.float64) / 5.0
    w3 = (w1[:, None, None] * w1[None, :, None] * w1[None, None, :])[None, None]
    pre = torch.nn.functional.conv3d(d.double()[None, None], w3, padding=1)[0, 0]
    ref = pre.clamp_min(0).float()
    gref = torch.nn.functional.conv3d((go.double() * (pre >= 0))[None, None], w3, padding=1)[0, 0].float()
    line = "%-16s fwd %s bwd %s  |fwd-ref| %.2e |bwd-ref| %.2e" % ("x".join(map(str, shape)), digest(out), digest(gd),
                                                             (out - ref).abs().max().item(), (gd - gref).abs().max().item())
    for name, f in (("fwd", lambda: ops.smooth3d_relu_fwd(d, 3.0, out=out)), ("bwd", lambda: ops.smooth3d_relu_bwd(out, go, 3.0, g_d=gd))):
        f(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        line += "  %s %.1f us" % (name, e0.elapsed_time(e1) * 20)
    print(line)
