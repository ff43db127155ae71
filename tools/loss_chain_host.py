"""one-view loss chains (configs[4] on VGG, the reference smokegun configuration on Inception-v1): GPU time per call, the
time the host needs to issue it, and the same call replayed as a hipGraph"""
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_flow_style_amd import engine, vgg, inception, synthetic as S
dev = "cuda:0"
rng = np.random.RandomState(0)
def measure(name, loss, d):
    g = torch.zeros_like(d)
    def f():
        g.zero_(); return loss.loss_and_grad(d, None, g)
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(50): f()
    e1.record(); host = (time.perf_counter() - t0) / 50 * 1e3
    torch.cuda.synchronize()
    print("%s: %.3f ms per call, host issue %.3f ms" % (name, e0.elapsed_time(e1) / 50, host))
    # graph
    gph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(gph):
        out = f()
    for _ in range(5): gph.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50): gph.replay()
    e1.record(); torch.cuda.synchronize()
    print("%s as a hipGraph: %.3f ms per replay" % (name, e0.elapsed_time(e1) / 50))
net = vgg.VGG(vgg.synthetic_weights(123, upto="conv4_1"), dev)
l = engine.RenderStyleLoss(net, ["conv1_1", "conv2_1", "conv3_1", "conv4_1"], [1.0] * 4, 1.0, transmit=0.2, render_liquid=True, rotate=False)
l.set_style_image(S.style_image(200, 200, rng))
measure("configs[4] loss chain (VGG, 200^3, 1 view)", l, torch.tensor(S.blob_density(200, rng), device=dev))
net2 = inception.InceptionV1(inception.synthetic_weights(123, upto="mixed4b"), dev)
l2 = engine.RenderStyleLoss(net2, ["conv2d2", "mixed3b", "mixed4b"], [1.0] * 3, 1.0, transmit=0.01, resize_scale=1.5, rotate=False)
l2.set_style_image(S.style_image(300, 450, rng))
dd = torch.nn.functional.pad(torch.tensor(S.blob_density(200, rng), device=dev), (0, 0, 50, 50)).contiguous()
measure("reference smokegun config loss chain (Inception, 200x300x200)", l2, dd)
