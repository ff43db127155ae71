"""The batched Winograd GEMM of ONE layer shape under a forced 16-row tile (NFS_GEMM_RB=3 NFS_GEMM_BM=.. NFS_GEMM_BN=..,
read once per process) or the tuner's own choice: in-library event pairs around the GEMM launches of a conv call.
    python tools/gemm_tile_force.py [HW=25] [Ci=512] [Co=512] [B=8]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neural_flow_style_amd.ops as ops
from neural_flow_style_amd import _lib

HW, Ci, Co, B = [int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 25), (2, 512), (3, 512), (4, 8))]
L = _lib.lib()
x = torch.randn(B, HW, HW, Ci, device="cuda")
w = torch.randn(3, 3, Ci, Co, device="cuda") * 0.05
wf = ops.conv3x3_pack(w, 0)
out = torch.empty(B, HW, HW, Co, device="cuda")
fn = lambda: ops.conv3x3_fwd(x, wf, torch.zeros(Co, device="cuda"), Co, True, out=out)
for _ in range(3): fn()
torch.cuda.synchronize()
L.nfs_gemm_timer(1)
for _ in range(20): fn()
ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
L.nfs_gemm_timer_read(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n))
L.nfs_gemm_timer(0)
g = ms.value / max(n.value, 1)
print("BM=%s BN=%s  %dx%d %d->%d B=%d  gemm %.1f us  %.1f TF/s executed" % (
    os.environ.get("NFS_GEMM_BM", "tuned"), os.environ.get("NFS_GEMM_BN", "tuned"), HW, HW, Ci, Co, B, 1e3 * g,
    fl.value / max(n.value, 1) / g / 1e9))
