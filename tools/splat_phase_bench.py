import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import neural_flow_style_amd.ops as ops
from neural_flow_style_amd import synthetic as S, transform as T
N, G = 500000, 200
rng = np.random.RandomState(0)
p = torch.tensor(S.blob_particles(N, rng), device="cuda")
p = p[T.grid_order(p, [G, G, G])].contiguous()
cfg = ops.make_splat_cfg(3, [G, G, G], [G, G, G], 0.5, 4, 1000.0, 1, False, 0)
grid = torch.zeros(G, G, G, 1, device="cuda")
from neural_flow_style_amd import _lib
import ctypes as C
def f():
    _lib.call("nfs_p2g_fwd", ops._ptr(p), None, None, ops._ptr(grid), None, N, 1, C.byref(cfg), ops._stream())
f(); torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print("NFS_SPLAT_LDS=%s (1 default: LDS accumulation; 0: global atomics) p2g forward kernel alone, 5e5 particles in grid order: %.1f us" % (os.environ.get("NFS_SPLAT_LDS"), 1e3 * e0.elapsed_time(e1) / 20))
