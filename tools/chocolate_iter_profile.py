"""BASELINE configs[4] at the chocolate scale (5e5 particles -> 200^3, 'p' field, liquid render, one view, VGG-19
conv1_1..conv4_1, TF-Adam on the displacements) as a loop of whole iterations: run under
``rocprofv3 --kernel-trace --stats`` for the per-kernel table (NFS_GRAPH=0 keeps the loss chain eager, i.e. visible)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_flow_style_amd import engine, synthetic as S, transform as T
from neural_flow_style_amd.config import get_config
from neural_flow_style_amd.styler_3p import Styler
N, G = 500000, 200
dev = "cuda:0"
rng = np.random.RandomState(0)
cfg, _ = get_config([])
for k_, v_ in dict(network="vgg_19.ckpt", data_dir="/nonexistent", synthetic_weights=True, resolution=[G, G, G],
                   domain=[12.8] * 3, radius=0.025, support=4, nsize=1, rest_density=1000, k=3, clip=False,
                   target_field="p", num_frames=1, batch_size=1, frames_per_opt=120, window_sigma=9, interp=1,
                   lr=0.002, iter=1, octave_n=1, style_layer=["conv1_1", "conv2_1", "conv3_1", "conv4_1"],
                   w_style_layer=[1.0] * 4, w_style=1.0, w_content=0, transmit=0.2, render_liquid=True, rotate=False,
                   resize_scale=1.0, num_kernels=1, kernel_scale=2, style_target=S.style_image(G, G, rng)).items():
    setattr(cfg, k_, v_)
cfg.rng = np.random.RandomState(123)
stp = Styler(cfg)
stp.load_img([G, G])
stp.loss.set_style_image(stp._style_feature(stp.style_img, [G, G]))
pp = torch.tensor(S.blob_particles(N, rng), device=dev)
pp = pp[T.grid_order(pp, [G, G, G])].contiguous()
var = torch.zeros(N, 3, device=dev)
adam = engine.TFAdamState()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 100):
    losses, g = stp._value_and_grad(pp, None, var, [G, G, G], stp._identity)
    adam.step(var, g.contiguous(), cfg.lr)
torch.cuda.synchronize()
