"""Per-wave phase cycle sums of winograd_fused_kernel (needs a library built with `make ABLATE=1`: nfs_fused_prof).
Phases are delimited by s_memtime reads at ISSUE time (indicative split; the robust figure is cycles per slice
against 72 MFMAs x 32 = 2304 cycles of matrix-pipe work)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import neural_flow_style_amd.ops as ops
from neural_flow_style_amd import _lib
L = _lib.lib()
L.nfs_fused_prof.argtypes = [ctypes.c_void_p]
for (HW, Ci, Co, kind) in [(200, 64, 64, "fwd"), (100, 64, 128, "fwd"), (100, 128, 64, "fwd"), (100, 128, 128, "fwd"),
                           (200, 64, 64, "dgrad_pool"), (100, 128, 128, "dgrad_pool")]:
    B = 8
    x = torch.relu(torch.randn(B, HW, HW, Ci, device="cuda")); w = torch.randn(3, 3, Ci, Co, device="cuda") * 0.05
    b = torch.zeros(Co, device="cuda"); wf = ops.conv3x3_pack(w, 0); out = torch.empty(B, HW, HW, Co, device="cuda")
    if kind == "fwd":
        fn = lambda: ops.conv3x3_fwd(x, wf, b, Co, True, out=out)
    else:
        wd = ops.conv3x3_pack(w, 1)
        bits = ops.conv3x3_relu_bits(B, HW, HW, Ci, Co, True, x.device)
        ops.conv3x3_fwd_pool(x, wf, b, Co, relu=True, relu_bits=bits, want_y=False)
        gyp = torch.randn(B, HW // 2, HW // 2, Co, device="cuda"); add = torch.randn(B, HW, HW, Ci, device="cuda")
        fn = lambda: ops.conv3x3_dgrad_pool(gyp, None, wd, Ci, x_in=x, addend=add, relu_bits=bits, hw=(HW, HW),
                                            addend_unmasked=True)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    buf = torch.zeros(8 * 4 * 16384, dtype=torch.int64, device="cuda")
    L.nfs_fused_prof(buf.data_ptr()); fn(); torch.cuda.synchronize(); L.nfs_fused_prof(None)
    p = buf.cpu().numpy().reshape(-1, 8)
    p = p[p[:, 7] > 0]
    ph = p[:, :5].astype(np.float64)
    tot = p[:, 5].astype(np.float64)
    t0 = p[:, 6].min()
    start = (p[:, 6] - t0) / 100.0; end = (p[:, 7] - t0) / 100.0      # us (s_memrealtime: 100 MHz)
    span = end.max()
    halves = Ci // 16
    print(kind, "%dx%d %d->%d: %d waves, wave life %.0f cycles avg, kernel span %.1f us, %d slices" % (HW, HW, Ci, Co, len(p), tot.mean(), span, halves))
    late = start > 0.25 * span
    print("   timeline: %d waves start in the first quarter (life %.1f us avg, last ends %.1f us); %d start later "
          "(at %.1f..%.1f us, life %.1f us avg)" % ((~late).sum(), (end - start)[~late].mean(), end[~late].max(), late.sum(),
          start[late].min() if late.any() else 0, start[late].max() if late.any() else 0, (end - start)[late].mean() if late.any() else 0))
    print("   waves running over time (10 bins):", [int(((start <= x) & (end > x)).sum()) for x in np.linspace(0, span, 11)[:-1] + span / 20])
    names = ["first slice staged+transformed", "fetch issue + MFMAs", "stash + barrier", "transform + barrier", "epilogue"]
    for n, v in zip(names, ph.mean(0)):
        print("   %-24s %8.0f  (%4.1f %%)  per slice %7.0f" % (n, v, 100 * v / tot.mean(), v / halves))
