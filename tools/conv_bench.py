"""Per-layer timing of the VGG conv kernels (fwd + dgrad) at the benchmark shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neural_flow_style_amd.ops as ops

LAYERS = [(200, 3, 64), (200, 64, 64), (100, 64, 128), (100, 128, 128), (50, 128, 256), (50, 256, 256),
          (25, 256, 512), (25, 512, 512), (12, 512, 512)]


def bench(B, reps=5):
    tot_f = tot_t = 0.0
    for HW, Ci, Co in LAYERS:
        x = torch.randn(B, HW, HW, Ci, device="cuda")
        w = torch.randn(3, 3, Ci, Co, device="cuda") * 0.05
        b = torch.zeros(Co, device="cuda")
        wf, wd = ops.conv3x3_pack(w, 0), ops.conv3x3_pack(w, 1)
        gy = torch.randn(B, HW, HW, Co, device="cuda")
        out = torch.empty(B, HW, HW, Co, device="cuda"); gx = torch.empty(B, HW, HW, Ci, device="cuda")
        res = []
        for fn in (lambda: ops.conv3x3_fwd(x, wf, b, Co, True, out=out),
                   lambda: ops.conv3x3_dgrad(gy, wd, Ci, x_in=(x if (Ci != 3 and not os.environ.get("NO_XIN")) else None), out=gx)):
            fn(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / reps)
        fl = 2.0 * B * HW * HW * 9 * Ci * Co
        tot_f += 2 * fl; tot_t += sum(res)
        print("B=%d %3dx%-3d %3d->%-3d  fwd %7.3f ms %6.1f TF/s   dgrad %7.3f ms %6.1f TF/s" %
              (B, HW, HW, Ci, Co, res[0], fl / res[0] / 1e9, res[1], fl / res[1] / 1e9))
    print("B=%d total %.3f ms  %.1f TF/s" % (B, tot_t, tot_f / tot_t / 1e9))


if __name__ == "__main__":
    for B in [int(a) for a in sys.argv[1:]] or (8, 1):
        bench(B)
