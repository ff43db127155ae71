"""Inception-v1 loss network (csrc/inception.hip) on its own: forward + data gradient of the reference driver's
configuration (test_smokegun.py:128-148: one 300 x 450 image = the 200 x 300 render resized by 1.5, tensors down to
mixed4b) and of 8 views of 200 x 200, with the per-call table of the C-ABI entry points (HIP events on the launch
stream) and the executed MFMA flops of the convolution kernel.

    python tools/inception_bench.py [--views 1 --h 300 --w 450] [--upto mixed4b] [--steps 20]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_flow_style_amd import _lib, inception, ops  # noqa: E402


def conv_flops(args):
    """executed MFMA flops of one nfs_conv2d_fwd call (K rounded up to whole 16-wide chunks, N to 64, M to the tile)"""
    (x, ldx, xm, ldm, wp, bias, y, ldy, yp, ldp, B, H, W, Cin, Cout, kh, kw, stride) = args[:18]
    Ho, Wo = ops.same_out(H, kh, stride)[0], ops.same_out(W, kw, stride)[0]
    M = B * Ho * Wo
    K = (kh * kw + 3) // 4 * 16 if Cin <= 4 else kh * kw * ((Cin + 15) // 16 * 16)
    N = (Cout + 63) // 64 * 64
    useful = 2.0 * M * Cout * kh * kw * Cin
    return 2.0 * M * N * K, useful


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=1)
    ap.add_argument("--h", type=int, default=300)
    ap.add_argument("--w", type=int, default=450)
    ap.add_argument("--upto", default="mixed4b")
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    dev = "cuda:0"
    net = inception.InceptionV1(inception.synthetic_weights(123, upto=inception.unit_of(a.upto)), dev)
    layers = ["conv2d2", "mixed3b", "mixed4b"] if a.upto == "mixed4b" else [a.upto]
    x = torch.randn(a.views, a.h, a.w, 3, device=dev) * 60

    def step():
        acts = net.forward(x, a.upto, keep=set(layers))
        grads = {n: acts[n] * 1e-3 for n in layers}
        return net.backward(acts, grads, a.upto)

    import time
    for grouped in (False, True):
        inception.GROUP_MAX_PIXELS = (1 << 16) if grouped else 0
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(a.steps):
            step()
        e1.record()
        host = (time.perf_counter() - t0) / a.steps * 1e3
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.steps
        print("forward + data gradient, %d x %d x %d down to %s, %s: %.3f ms (host needs %.3f ms to issue it)"
              % (a.views, a.h, a.w, a.upto, "grouped module launches" if grouped else "one launch per branch", ms, host))

    _lib.PROFILE = {}
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    prof, _lib.PROFILE = _lib.PROFILE, None
    gshapes = prof.pop("shapes:nfs_conv2d_group", [])
    tot_ms, conv_ms, conv_exec, conv_useful = 0.0, 0.0, 0.0, 0.0
    print("%-26s %6s %9s %9s" % ("entry point", "calls", "ms/step", "TFLOP/s"))
    for name, recs in sorted(prof.items(), key=lambda kv: -sum(r[0].elapsed_time(r[1]) for r in kv[1])):
        t = sum(r[0].elapsed_time(r[1]) for r in recs) / 5
        line = "%-26s %6d %9.3f" % (name, len(recs) // 5, t)
        if name in ("nfs_conv2d_fwd", "nfs_conv2d_group"):
            if name == "nfs_conv2d_fwd":
                fl = [conv_flops(r[2]) for r in recs]
            else:
                fl = [conv_flops((0,) * 10 + sh + (1,)) for call in gshapes for sh in call]
            ex, us = sum(f[0] for f in fl) / 5, sum(f[1] for f in fl) / 5
            line += " %9.1f executed (%.1f on the direct-convolution flops; %.2f GF / %.2f GF per step)" % (
                ex / t / 1e9, us / t / 1e9, ex / 1e9, us / 1e9)
            conv_ms, conv_exec, conv_useful = conv_ms + t, conv_exec + ex, conv_useful + us
        tot_ms += t
        print(line)
    print("sum of the calls %.3f ms (event pairs add ~7 us each); convolutions %.3f ms, %.1f TFLOP/s executed = %.1f%% of the "
          "f32 MFMA peak (%.1f TFLOP/s on the direct-convolution flops)"
          % (tot_ms, conv_ms, conv_exec / conv_ms / 1e9, 100 * conv_exec / conv_ms / 1e9 / 157.3, conv_useful / conv_ms / 1e9))
    # the slowest convolution calls
    recs = prof["nfs_conv2d_fwd"]
    per = {}
    for r in recs:
        k = tuple(r[2][10:18]) + (bool(r[2][2]),)
        per.setdefault(k, []).append(r[0].elapsed_time(r[1]))
    rows = sorted(((np.mean(v) * len(v) / 5, k, np.mean(v)) for k, v in per.items()), reverse=True)[:12]
    print("slowest convolution shapes (B,H,W,Cin,Cout,kh,kw,stride,dgrad): ms/step total, us per call, TFLOP/s executed")
    for tot, k, mean in rows:
        ex, _ = conv_flops((0,) * 10 + k[:8])
        print("  %-44s %7.3f %8.1f %7.1f" % (k, tot, mean * 1e3, ex / mean / 1e9))


if __name__ == "__main__":
    main()
