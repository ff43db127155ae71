"""host cost of C-ABI calls (no synchronisation inside the timed loops; short loops so that the launch queue never fills)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_flow_style_amd import ops, _lib, vgg
dev = "cuda:0"
def host(f, n=200):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return dt
x = torch.zeros(1024, device=dev)
L = _lib.lib()
s = ops._stream()
print("raw ctypes nfs_fill (1 launch): %.1f us" % host(lambda: L.nfs_fill(x.data_ptr(), 0.0, 1024, s)))
print("ops.fill: %.1f us" % host(lambda: ops.fill(x, 0.0)))
print("torch zero_: %.1f us" % host(lambda: x.zero_()))
w = vgg.synthetic_weights(123, upto="conv4_2")
net = vgg.VGG(w, dev)
for name, hw, B in (("conv3_2", 50, 8), ("conv4_2", 25, 8), ("conv1_2", 200, 8)):
    p = net.params[name]
    a = torch.randn(B, hw, hw, p["cin"], device=dev)
    f = lambda: ops.conv3x3_fwd(a, p["fwd"], p["bias"], p["cout"], relu=True)
    for _ in range(5): f()
    print("ops.conv3x3_fwd %s (B=%d, %dx%d): %.1f us" % (name, B, hw, hw, host(f, 100)))
