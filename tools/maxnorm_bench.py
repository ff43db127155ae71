"""max-normalisation + loss-net input in one launch (nfs_maxnorm_input_fwd) and its adjoint at the headline shape."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neural_flow_style_amd.ops as ops
V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
img = torch.rand(V, 200, 200, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) + 0.1
def timed(f, reps=50):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3
x, gm = ops.maxnorm_input_fwd(img, V)
t = timed(lambda: ops.maxnorm_input_fwd(img, V))
gx = torch.randn(V, 200, 200, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
tb = timed(lambda: ops.maxnorm_input_bwd(img, gm, gx))
print("V=%d maxnorm_input fwd %.1f us  bwd %.1f us  digest %s" % (V, t, tb, hashlib.sha1(x.cpu().numpy().tobytes()).hexdigest()[:12]))
