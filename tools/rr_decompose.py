import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import neural_flow_style_amd.ops as ops
from neural_flow_style_amd import synthetic as S, transform as T
G = 200
V = 8
d = torch.tensor(S.blob_density(G, np.random.RandomState(0)), device="cuda")
rot = T.rot_to_device(S.uniform_views(V), "cuda")
rot0 = T.rot_to_device([np.eye(3)] * V, "cuda")
d_rot = torch.empty(V, G, G, G, device="cuda")
def t(f):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(40): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 40)
    return best
print("reuse", os.environ.get("NFS_RR_REUSE"), "store %.3f  nostore %.3f  identity-rot store %.3f nostore %.3f  memset256MB %.3f" % (
    t(lambda: ops.rotate_render_fwd(d, rot, 0.01, False, d_rot=d_rot)),
    t(lambda: ops.rotate_render_fwd(d, rot, 0.01, False)),
    t(lambda: ops.rotate_render_fwd(d, rot0, 0.01, False, d_rot=d_rot)),
    t(lambda: ops.rotate_render_fwd(d, rot0, 0.01, False)),
    t(lambda: d_rot.fill_(1.0))))
