import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import neural_flow_style_amd.ops as ops
from neural_flow_style_amd import synthetic as S, transform as T
G, V = 200, 8
g = torch.randn(V, G, G, G, 1, device="cuda")
rot = T.rot_to_device(S.uniform_views(V), "cuda")
acc = torch.zeros(G, G, G, 1, device="cuda")
for _ in range(2): ops.rotate_bwd(g, rot, g_d_acc=acc)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): ops.rotate_bwd(g, rot, g_d_acc=acc)
e1.record(); torch.cuda.synchronize()
print("variant", os.environ.get("NFS_RT_VARIANT", "0"), "rotate_bwd ms", e0.elapsed_time(e1) / 5)
