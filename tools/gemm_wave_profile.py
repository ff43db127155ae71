"""Per-wave phase cycle sums of winograd_gemm_kernel (needs a library built with `make ABLATE=1`: nfs_gemm_prof).
Phases are delimited by s_memtime reads at ISSUE time; MFMA issue blocks while the SIMD partner\x27s MFMAs occupy the pipe,
so the split between phases is indicative only -- the robust figure is cycles per chunk against 2 x 4096 (two waves share
a SIMD\x27s matrix pipe): conv3_x 10034 -> 82 % pipe use inside the K loop."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import neural_flow_style_amd.ops as ops
from neural_flow_style_amd import _lib
L = _lib.lib()
L.nfs_gemm_prof.argtypes = [ctypes.c_void_p]
for (HW, Ci, Co) in [(50, 256, 256), (25, 512, 512), (100, 128, 128)]:
    B = 8
    x = torch.randn(B, HW, HW, Ci, device="cuda"); w = torch.randn(3, 3, Ci, Co, device="cuda") * 0.05
    b = torch.zeros(Co, device="cuda"); wf = ops.conv3x3_pack(w, 0); out = torch.empty(B, HW, HW, Co, device="cuda")
    fn = lambda: ops.conv3x3_fwd(x, wf, b, Co, True, out=out)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    buf = torch.zeros(8 * 4 * 8192, dtype=torch.int64, device="cuda")
    L.nfs_gemm_prof(buf.data_ptr()); fn(); torch.cuda.synchronize(); L.nfs_gemm_prof(None)
    p = buf.cpu().numpy().reshape(-1, 8)
    p = p[p[:, 7] > 0]
    ph = p[:, :6].astype(np.float64)
    tot = (p[:, 7] - p[:, 6]).astype(np.float64)
    span = (p[:, 7].max() - p[:, 6].min())
    print("%dx%d %d->%d: %d waves, wave life %.0f cycles avg, kernel span %.0f ticks" % (HW, HW, Ci, Co, len(p), tot.mean(), span))
    names = ["load+stage wait", "barrier", "load issue", "frag+MFMA", "drain", "epilogue"]
    for n, v in zip(names, ph.mean(0)):
        print("   %-16s %8.0f  (%4.1f %%)" % (n, v, 100 * v / tot.mean()))
    print("   prologue (before loop) %.0f" % (tot.mean() - ph.sum(1).mean()))
