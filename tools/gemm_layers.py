"""Per-layer time of the batched Winograd GEMM alone (in-library HIP event pairs) next to the whole conv call."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neural_flow_style_amd.ops as ops
from neural_flow_style_amd import _lib

LAYERS = [(200, 64, 64), (100, 64, 128), (100, 128, 128), (50, 128, 256), (50, 256, 256),
          (25, 256, 512), (25, 512, 512), (12, 512, 512)]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = _lib.lib()
tot = 0.0
for HW, Ci, Co in LAYERS:
    x = torch.randn(B, HW, HW, Ci, device="cuda")
    w = torch.randn(3, 3, Ci, Co, device="cuda") * 0.05
    b = torch.zeros(Co, device="cuda")
    wf = ops.conv3x3_pack(w, 0)
    out = torch.empty(B, HW, HW, Co, device="cuda")
    fn = lambda: ops.conv3x3_fwd(x, wf, b, Co, True, out=out)
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    whole = e0.elapsed_time(e1) / 10
    L.nfs_gemm_timer(1)
    for _ in range(10): fn()
    ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
    L.nfs_gemm_timer_read(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n))
    L.nfs_gemm_timer(0)
    T = B * ((HW + 3) // 4) ** 2
    if n.value == 0:                      # narrow layer: the single-kernel path, no batched GEMM launch
        print("B=%d %3dx%-3d %3d->%-3d T=%5d  conv %.3f ms  (winograd_fused_kernel)" % (B, HW, HW, Ci, Co, T, whole))
        whole_tot = whole_tot + whole if "whole_tot" in dir() else whole
        continue
    g = ms.value / n.value
    whole_tot = whole_tot + whole if "whole_tot" in dir() else whole
    tot += g
    print("B=%d %3dx%-3d %3d->%-3d T=%5d  conv %.3f ms  gemm %.4f ms %6.1f TF/s executed  (transforms %.3f)" %
          (B, HW, HW, Ci, Co, T, whole, g, fl.value / n.value / g / 1e9, whole - g))
print("gemm total %.3f ms, conv calls total %.3f ms" % (tot, whole_tot))
