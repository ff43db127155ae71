"""Narrow VGG layers (<= 128 channels either side) through nfs_conv3x3_*: time per pass, and -- with --compare -- the
single-kernel Winograd path (winograd_fused.hip) against the three-kernel one (NFS_WG_FUSED=0) on the same operands.

    python tools/fused_conv_bench.py            # times the path the library takes
    python tools/fused_conv_bench.py --compare  # runs itself twice (fused / NFS_WG_FUSED=0), diffs outputs and times
"""
import argparse
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# the headline configuration's narrow layers: (name, H, Ci, Co, pooled)
LAYERS = [("conv1_2", 200, 64, 64, True), ("conv2_1", 100, 64, 128, False), ("conv2_2", 100, 128, 128, True)]


def run(out_path, B, reps):
    import torch
    import neural_flow_style_amd  # noqa: F401  (registers the package under its importable name)
    from neural_flow_style_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    res, dump = {}, {}

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    for name, H, Ci, Co, pooled in LAYERS:
        w = (torch.randn(3, 3, Ci, Co, generator=g) * (2.0 / (9 * Ci)) ** 0.5).to(dev)
        bias = (0.01 * torch.randn(Co, generator=g)).to(dev)
        x = torch.relu(torch.randn(B, H, H, Ci, generator=g)).to(dev)
        pf, pd = ops.conv3x3_pack(w, 0), ops.conv3x3_pack(w, 1)
        bits = ops.conv3x3_relu_bits(B, H, H, Ci, Co, pooled, dev)
        # forward
        if pooled:
            y, yp = ops.conv3x3_fwd_pool(x, pf, bias, Co, relu=True, relu_bits=bits, want_y=True)
            res[name + " fwd+pool"] = timed(lambda: ops.conv3x3_fwd_pool(x, pf, bias, Co, relu=True, relu_bits=bits,
                                                                        want_y=False))
            dump[name + ".y"], dump[name + ".yp"] = y, yp
        else:
            y = ops.conv3x3_fwd(x, pf, bias, Co, relu=True, relu_bits=bits)
            res[name + " fwd"] = timed(lambda: ops.conv3x3_fwd(x, pf, bias, Co, relu=True, relu_bits=bits))
            dump[name + ".y"] = y
        # data gradient (mask of x from the bit cache, an addend that has not been through it)
        add = torch.randn(B, H, H, Ci, generator=g).to(dev)
        if pooled:
            gyp = torch.randn(B, H // 2, H // 2, Co, generator=g).to(dev)
            fn = lambda: ops.conv3x3_dgrad_pool(gyp, None, pd, Ci, x_in=x, addend=add, relu_bits=bits, hw=(H, H),
                                                addend_unmasked=True)
            res[name + " dgrad(pool)"] = timed(fn)
            dump[name + ".gx"] = fn()
        else:
            gy = torch.randn(B, H, H, Co, generator=g).to(dev)
            fn = lambda: ops.conv3x3_dgrad(gy, pd, Ci, x_in=None, addend=None)
            res[name + " dgrad"] = timed(fn)
            dump[name + ".gx"] = fn()
            fn2 = lambda: ops.conv3x3_dgrad(gy, pd, Ci, x_in=x, addend=add, relu_bits=bits)
            dump[name + ".gx_masked"] = fn2()
        dump[name + ".bits"] = bits.clone() if bits is not None else torch.zeros(1)
        if bits is not None:
            # the mask of x the forward pass recorded, against a host computation (word = 4x4 tile x channel pair)
            xb = (x > 0).cpu().numpy().reshape(B, H // 4, 4, H // 4, 4, Ci // 2, 2)
            sh = (np.arange(4)[:, None, None] * 4 + np.arange(4)[None, :, None]) * 2 + np.arange(2)[None, None, :]
            ref = (xb.transpose(0, 1, 3, 5, 2, 4, 6).astype(np.uint64) << sh.astype(np.uint64)).sum((4, 5, 6))
            got = bits.cpu().numpy().view(np.uint32)[:ref.size].reshape(ref.shape)
            res[name + " in_bits wrong words"] = int((got != ref.astype(np.uint32)).sum())
    torch.cuda.synchronize()
    np.savez(out_path, **{k: v.detach().cpu().numpy() for k, v in dump.items()})
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--compare", action="store_true")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    if not args.compare:
        res = run(args.out or "/tmp/fused_conv.npz", args.batch, args.reps)
        print(json.dumps(res))
        return
    outs = {}
    for tag, env in (("fused", {}), ("three_kernel", {"NFS_WG_FUSED": "0"})):
        path = "/tmp/fused_conv_%s.npz" % tag
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--batch", str(args.batch), "--reps",
                            str(args.reps), "--out", path], env=dict(os.environ, **env), capture_output=True, text=True)
        if p.returncode:
            print(p.stdout[-2000:], p.stderr[-4000:])
            sys.exit(1)
        outs[tag] = (json.loads(p.stdout.strip().splitlines()[-1]), np.load(path))
    ta, a = outs["fused"]
    tb, b = outs["three_kernel"]
    print("%-24s %10s %10s" % ("pass", "fused ms", "3-kernel ms"))
    for k in ta:
        print("%-24s %10.4f %10.4f" % (k, ta[k], tb[k]))
    print("total %.4f vs %.4f ms" % (sum(v for k, v in ta.items() if "wrong" not in k),
                                     sum(v for k, v in tb.items() if "wrong" not in k)))
    worst = 0.0
    for k in a.files:
        if k.endswith(".bits"):
            ua, ub = a[k].view(np.uint32), b[k].view(np.uint32)      # (the cache travels in a float32 tensor)
            diff = int((ua != ub).sum())
            flips = int(sum(bin(int(v)).count("1") for v in (ua ^ ub)[ua != ub]))
            print("%-24s differing words %d of %d (%d mask bits: outputs within rounding of zero)" % (k, diff, ua.size, flips))
            continue
        den = float(np.linalg.norm(b[k].astype(np.float64)))
        rel = float(np.linalg.norm(a[k].astype(np.float64) - b[k])) / max(den, 1e-30)
        worst = max(worst, rel)
        print("%-24s rel L2 %.3e" % (k, rel))
    print("worst rel L2 %.3e" % worst)


if __name__ == "__main__":
    main()
