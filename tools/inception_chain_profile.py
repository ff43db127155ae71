"""the reference driver's own configuration (Inception-v1 conv2d2 / mixed3b / mixed4b on the 300 x 450 render of a
200 x 300 x 200 field, one unrotated view) as a loop of loss + gradient calls: run under
``rocprofv3 --kernel-trace --stats`` for the per-kernel table of that chain"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_flow_style_amd import engine, inception, synthetic as S
dev = "cuda:0"
rng = np.random.RandomState(3)
net = inception.InceptionV1(inception.synthetic_weights(123, upto="mixed4b"), dev)
il = engine.RenderStyleLoss(net, ["conv2d2", "mixed3b", "mixed4b"], [1.0] * 3, 1.0, transmit=0.01, resize_scale=1.5, rotate=False)
il.set_style_image(S.style_image(300, 450, rng))
dd = torch.nn.functional.pad(torch.tensor(S.blob_density(200, rng), device=dev), (0, 0, 50, 50)).contiguous()
g = torch.zeros_like(dd)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 100):
    g.zero_()
    il.loss_and_grad(dd, None, g)
torch.cuda.synchronize()
