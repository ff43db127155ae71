"""the deep VGG layers at the headline's 8 views, forward + data gradient, with the Winograd GEMMs on the f32-input MFMA
(mode 0) and in split-limb arithmetic on the bf16 MFMA (mode 1: the 16-row register-B instance rb16s): call times
(input transform + GEMM + output transform) and the GEMM alone (the library's own event pairs, nfs_gemm_timer).
    python tools/split_gemm_bench.py [views=8]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neural_flow_style_amd.ops as ops
from neural_flow_style_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = _lib.lib()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e3
    L.nfs_gemm_timer(1)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
    L.nfs_gemm_timer_read(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n))
    L.nfs_gemm_timer(0)
    return t, 1e3 * ms.value / max(n.value, 1), fl.value / max(ms.value, 1e-9) / 1e9


layers = [("conv3_1", 50, 128, 256, False), ("conv3_2", 50, 256, 256, False), ("conv3_4", 50, 256, 256, True),
          ("conv4_1", 25, 256, 512, False), ("conv4_2", 25, 512, 512, False), ("conv4_4", 25, 512, 512, True),
          ("conv5_1", 12, 512, 512, False)]
tot = {0: [0.0, 0.0], 1: [0.0, 0.0]}
for name, HW, Ci, Co, pooled in layers:
    x = torch.relu(torch.randn(B, HW, HW, Ci, device="cuda")); w = torch.randn(3, 3, Ci, Co, device="cuda") * 0.03
    b = torch.zeros(Co, device="cuda"); wf, wd = ops.conv3x3_pack(w, 0), ops.conv3x3_pack(w, 1)
    gy = torch.randn(B, HW, HW, Co, device="cuda"); add = torch.randn(B, HW, HW, Ci, device="cuda")
    row = [name]
    for mode in (0, 1):
        ops.gemm_mode(mode)
        if pooled:
            f = lambda: ops.conv3x3_fwd_pool(x, wf, b, Co, relu=True)
        else:
            f = lambda: ops.conv3x3_fwd(x, wf, b, Co, True)
        g = lambda: ops.conv3x3_dgrad(gy, wd, Ci, x_in=x, addend=add)
        tf, gf, tff = timed(f)
        tb, gb, tfb = timed(g)
        tot[mode][0] += gf; tot[mode][1] += gb
        row += [tf, gf, tff, tb, gb, tfb]
    print("%-8s mode 0: fwd %6.1f us (GEMM %6.1f, %5.1f TF/s) dgrad %6.1f us (GEMM %6.1f, %5.1f TF/s) | mode 1: fwd %6.1f us "
          "(GEMM %6.1f, %5.1f TF/s) dgrad %6.1f us (GEMM %6.1f, %5.1f TF/s)" % tuple(row), flush=True)
ops.gemm_mode(0)
print("GEMM totals: mode 0 fwd %.1f dgrad %.1f us | mode 1 fwd %.1f dgrad %.1f us" % (tot[0][0], tot[0][1], tot[1][0], tot[1][1]))
