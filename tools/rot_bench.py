import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import neural_flow_style_amd.ops as ops
from neural_flow_style_amd import synthetic as S, transform as T
G, V = 200, 8
g = torch.randn(V, G, G, G, 1, device="cuda", generator=torch.Generator(device="cuda").manual_seed(11))
rot = T.rot_to_device(S.uniform_views(V), "cuda")
acc = torch.zeros(G, G, G, 1, device="cuda")
gmax = g.abs().max().reshape(1)
for _ in range(2): ops.rotate_bwd(g, rot, g_d_acc=acc, g_max=gmax)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(4):
    e0.record()
    for _ in range(10): ops.rotate_bwd(g, rot, g_d_acc=acc, g_max=gmax)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10)
print("float acc" if os.environ.get("NFS_RT_FLOAT") else "fixed point", "rotate_bwd ms %.4f" % best)
ref = torch.zeros_like(acc); ops.rotate_bwd(g, rot, g_d_acc=ref, g_max=gmax)
print("checksum %.9e" % float(ref.double().sum()), "l2 %.9e" % float(ref.double().norm()))
import hashlib
print("sha1", hashlib.sha1(ref.cpu().numpy().tobytes()).hexdigest()[:16])
