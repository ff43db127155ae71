"""Narrow VGG layers at sizes that are NOT multiples of 4 (the reference driver's 300 x 450 render; the lower octaves of
test_dambreak2d.py): the single-kernel Winograd path's ragged instance against the three-kernel path (NFS_WG_FUSED_RAG=0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_flow_style_amd import ops

def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(3)
tot = 0.0
for B, shapes in ((1, [("conv1_2", 300, 450, 64, 64, True), ("conv2_1", 150, 225, 64, 128, False), ("conv2_2", 150, 225, 128, 128, True)]),
                  (1, [("conv1_2", 301, 602, 64, 64, True), ("conv2_1", 150, 301, 64, 128, False)]),
                  (8, [("conv1_2", 150, 150, 64, 64, True), ("conv2_1", 75, 75, 64, 128, False), ("conv2_2", 75, 75, 128, 128, True)])):
    for name, H, W, Ci, Co, pooled in shapes:
        w = (torch.randn(3, 3, Ci, Co, generator=g) * (2.0 / (9 * Ci)) ** 0.5).to(dev)
        bias = (0.01 * torch.randn(Co, generator=g)).to(dev)
        x = torch.relu(torch.randn(B, H, W, Ci, generator=g)).to(dev)
        pf, pd = ops.conv3x3_pack(w, 0), ops.conv3x3_pack(w, 1)
        bits = ops.conv3x3_relu_bits(B, H, W, Ci, Co, pooled, dev)
        add = torch.randn(B, H, W, Ci, generator=g).to(dev)
        if pooled:
            ops.conv3x3_fwd_pool(x, pf, bias, Co, relu=True, relu_bits=bits, want_y=True)
            tf = timed(lambda: ops.conv3x3_fwd_pool(x, pf, bias, Co, relu=True, relu_bits=bits, want_y=False))
            gyp = torch.randn(B, H // 2, W // 2, Co, generator=g).to(dev)
            tb = timed(lambda: ops.conv3x3_dgrad_pool(gyp, None, pd, Ci, x_in=x, addend=add, relu_bits=bits, hw=(H, W), addend_unmasked=True))
        else:
            tf = timed(lambda: ops.conv3x3_fwd(x, pf, bias, Co, relu=True, relu_bits=bits))
            gy = torch.randn(B, H, W, Co, generator=g).to(dev)
            tb = timed(lambda: ops.conv3x3_dgrad(gy, pd, Ci, x_in=x, addend=add, relu_bits=bits))
        tot += tf + tb
        print("B %d %-8s %3d x %3d  %3d -> %3d%s: forward %.4f ms, data gradient %.4f ms" % (B, name, H, W, Ci, Co, " (+pool)" if pooled else "", tf, tb))
print("sum %.4f ms  [%s]" % (tot, "three-kernel path for ragged sizes" if os.environ.get("NFS_WG_FUSED_RAG") == "0" else "single-kernel ragged instance"))
