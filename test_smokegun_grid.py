"""Driver of the grid (TNST-style) path on the MI355X build -- BASELINE ``configs[3]``: a frame sequence on a 3-D smoke
grid stylised through ``styler_grid.Styler`` (per-frame stylisation velocity through ``advect``, updates aligned across
frames by ``_transport`` + the ``denoise`` Gaussian, frames sharded over the ranks of a ``torch.distributed`` launch).

The mounted reference branch has no driver for this path (its TNST code lives on a branch that is not mounted,
README.md:11); the I/O conventions are the ones its other drivers use: per frame ``d_path`` = npz key ``x`` density
[D,H,W] with H flipped on read (test_smokegun_resim.py:232-233), ``v_path`` = mantaflow MAC velocity, converted to
cell-centred (test_smokegun_resim.py:235-243); results as ``%03d.npz`` (key ``x`` = ``d[:, ::-1]``), ``%03d.png``,
``loss_plot.png`` (test_smokegun.py:81-101).  Without a dataset it runs on seeded synthetic frames.

    python test_smokegun_grid.py --style_target data/image/fire_new.jpg --w_style 1 --num_frames 8
    python -m torch.distributed.run --nproc-per-node 8 test_smokegun_grid.py --num_frames 60 ...
"""
import os

import numpy as np

from config import get_config
from styler_grid import Styler
from util import prepare_dirs_and_logger
from neural_flow_style_amd.resim import mac_to_centered


def to_advect_units(v_, scale=1.0):
    """cell-centred (x,y,z) velocity in cells per frame, H already flipped, [D,H,W,3] -> the units of ``advect``:
    component k along array axis k in normalised coordinates (one cell = 2/(n-1)); the y
    component changes sign with the H flip"""
    D, H, W = v_.shape[:3]
    return np.stack([2.0 * v_[..., 2] / max(D - 1, 1), -2.0 * v_[..., 1] / max(H - 1, 1),
                     2.0 * v_[..., 0] / max(W - 1, 1)], axis=-1).astype(np.float32) * np.float32(scale)


def load_frame(config, t):
    d_path = os.path.join(config.data_dir, config.dataset, config.d_path % (config.target_frame + t))
    v_path = os.path.join(config.data_dir, config.dataset, config.v_path % (config.target_frame + t))
    if not (os.path.exists(d_path) and os.path.exists(v_path)):
        return None
    with np.load(d_path) as data:
        d = np.ascontiguousarray(data["x"][:, ::-1], np.float32)
    with np.load(v_path) as data:
        u = to_advect_units(mac_to_centered(data["x"]))
    return d, u


def synthetic_frame(config, t):
    import test_smokegun_resim as R
    D, H, W = config.resolution
    d, _ = R.synthetic_frame(config, t)
    rng = np.random.RandomState(config.seed + 1000 + t)
    zz, yy, xx = np.meshgrid(np.linspace(0, 1, D), np.linspace(0, 1, H), np.linspace(0, 1, W), indexing="ij")
    mac = np.zeros((D, H, W, 3), np.float32)
    mac[..., 1] = 1.5
    mac[..., 0] = 0.5 * np.sin(2 * np.pi * zz)
    mac[..., 2] = 0.5 * np.cos(2 * np.pi * xx)
    mac += rng.randn(D, H, W, 3).astype(np.float32) * 0.05
    return d, to_advect_units(mac_to_centered(mac))


def run(config):
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    prepare_dirs_and_logger(config)
    config.rng = np.random.RandomState(config.seed)
    if not config.style_target:
        from neural_flow_style_amd import synthetic as S
        print("DEMO MODE: synthetic style image and synthetic (random) VGG-19 filters -- not a stylisation by VGG-19")
        config.style_target = S.style_image(256, 256, np.random.RandomState(config.seed))
        config.w_style = 1
        config.synthetic_weights = True
    styler = Styler(config)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        styler.pg = dist.group.WORLD
    styler.load_img(config.resolution[1:])
    frames = [load_frame(config, t) or synthetic_frame(config, t) for t in range(config.num_frames)]
    result = styler.run({"d": [f[0] for f in frames], "v": [f[1] for f in frames]})
    if rank != 0:
        return result
    from PIL import Image
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        plt.plot(range(len(result["l"][0])), result["l"][0], label="oct 0")
        plt.legend()
        plt.savefig(os.path.join(config.log_dir, "loss_plot.png"))
    except Exception as e:  # plotting is optional
        print("loss plot skipped:", e)
    for i, img in enumerate(result["r"]):
        Image.fromarray(img).save(os.path.join(config.log_dir, "%03d.png" % (config.target_frame + i)))
    for i, d in enumerate(result["d"]):
        np.savez_compressed(os.path.join(config.log_dir, "%03d.npz" % (config.target_frame + i)), x=d[:, ::-1])
    return result


def main(config):
    config.dataset = "smokegun"
    if config.resolution == [384, 288]:
        config.resolution = [200, 300, 200]
        config.resize_scale = 300 / config.resolution[0]
    config.k = 3
    config.batch_size = 1
    config.frames_per_opt = 1
    config.interp = 1
    config.octave_n = 1
    config.network = "vgg_19.ckpt"
    if config.style_layer == ["conv3_1"]:
        config.style_layer = ["conv1_1", "conv2_1", "conv3_1", "conv4_1", "conv5_1"]
        config.w_style_layer = [1, 1, 1, 1, 1]
    if not str(config.content_layer).startswith("conv"):
        config.w_content = 0
    config.transmit = 0.01
    config.rotate = True
    config.n_views = 8
    if not config.grid_variable:
        config.grid_variable = "v"
    if os.environ.get("WORLD_SIZE") and int(os.environ["WORLD_SIZE"]) > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("NFS_DIST_BACKEND", "nccl"))
    return run(config)


if __name__ == "__main__":
    cfg, _ = get_config()
    main(cfg)
