"""Per-layer timing of the Gram kernels (fwd two-pass, bwd GEMM) at the benchmark shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neural_flow_style_amd.ops as ops

LAYERS = [(200 * 200, 64), (100 * 100, 128), (50 * 50, 256), (25 * 25, 512), (12 * 12, 512)]
for B in [int(a) for a in sys.argv[1:]] or (8, 1):
    tf = tb = 0.0
    fl_all = 0.0
    for HW, C in LAYERS:
        F = torch.rand(B, HW, C, device="cuda")
        D = torch.randn(B, C, C, device="cuda"); D = D + D.transpose(1, 2)
        dF = torch.empty_like(F)
        res = []
        for fn in (lambda: ops.gram_fwd(F, 1.0 / (2 * HW * C)), lambda: ops.gram_bwd(F, D, 1.0, out=dF)):
            fn(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 10)
        fl = 2.0 * B * HW * C * C
        fl_all += fl; tf += res[0]; tb += res[1]
        print("B=%d HW=%6d C=%3d  fwd %7.3f ms %6.1f TF/s   bwd %7.3f ms %6.1f TF/s" %
              (B, HW, C, res[0], fl / res[0] / 1e9, res[1], fl / res[1] / 1e9))
    print("B=%d total fwd %.3f ms %.1f TF/s, bwd %.3f ms %.1f TF/s" % (B, tf, fl_all / tf / 1e9, tb, fl_all / tb / 1e9))
