"""Counterpart of the reference driver ``test_smokegun.py`` (test_smokegun.py:16-200) on the
MI355X build: same ``run(config)`` body -- build the Styler, load the style image, read the
particle frames, ``styler.run(params)``, save ``%03d.png`` / ``%03d.npz`` (key ``x`` = ``d[:, ::-1]``)
/ ``loss_plot.png`` -- with two differences: particles come from ``.npz`` files (keys ``position``
[N,3] in world units (x,y,z), ``density`` [N,num_kernels]) or partio ``.bgeo`` through ``io_bgeo``, and the
``main()`` override block yields to flags given on the command line (the reference hard-codes the Inception graph
and a single view; BASELINE's configuration is the VGG-19 style loss over 8 rotated views).  Without a dataset it
runs on seeded synthetic particles.

    python test_smokegun.py --style_target data/image/fire_new.jpg --w_style 1 --w_content 0          (run.bat:22)
    python test_smokegun.py --content_layer mixed3b_3x3_bottleneck_pre_relu --content_channel 44      (run.bat:14)
    python test_smokegun.py --network vgg_19.ckpt --rotate true --n_views 8 --w_style 1 --w_content 0 --style_target ...
"""
import os

import numpy as np

from config import get_config
from styler_3p import Styler
from util import prepare_dirs_and_logger


def load_frames(config):
    p, r = [], []
    nmax = 0
    raw = []
    for i in range(config.num_frames):
        path = os.path.join(config.data_dir, config.dataset, config.d_path % (config.target_frame + i))
        bgeo = os.path.splitext(path)[0] + ".bgeo"
        if path.endswith(".bgeo") or (not os.path.exists(path) and os.path.exists(bgeo)):
            # the reference's own format (test_smokegun.py:41-56): partio particle file; particle j reads the
            # position / density stored at index id[j]
            import io_bgeo as partio
            pt = partio.read(bgeo if not path.endswith(".bgeo") else path)
            ids = pt.array("id")[:, 0]
            raw.append((pt.array("position")[ids], pt.array("density")[ids]))
            nmax = max(nmax, raw[-1][0].shape[0])
            continue
        if not os.path.exists(path):
            if i == 0:
                return None                                    # no dataset at all: the caller's demo mode
            raise FileNotFoundError("frame %d of the sequence is missing: %s (frame 0 exists -- refusing to replace a "
                                    "partly present dataset by synthetic particles)" % (i, path))
        z = np.load(path)
        raw.append((np.asarray(z["position"], np.float32), np.asarray(z["density"], np.float32)))
        nmax = max(nmax, raw[-1][0].shape[0])
    for pos, den in raw:
        p_ = np.ones([nmax, 3], np.float32) * -1              # padded slots sit outside the domain
        r_ = np.zeros([nmax, config.num_kernels], np.float32)
        n = pos.shape[0]
        # normalise to [0,1] and order (z,y,x)  (test_smokegun.py:60-65)
        p_[:n] = np.stack([pos[:, 2] / config.domain[0], pos[:, 1] / config.domain[1], pos[:, 0] / config.domain[2]], -1)
        r_[:n] = den.reshape(n, -1)[:, :config.num_kernels]
        p.append(p_); r.append(r_)
    return {"p": p, "r": r}


def synthetic_frames(config):
    from neural_flow_style_amd import synthetic as S
    rng = np.random.RandomState(config.seed)
    n = 20000
    p = [S.blob_particles(n, rng) for _ in range(config.num_frames)]
    r = [rng.uniform(0.2, 1.0, (n, config.num_kernels)).astype(np.float32) for _ in range(config.num_frames)]
    return {"p": p, "r": r}


def run(config):
    prepare_dirs_and_logger(config)
    config.rng = np.random.RandomState(config.seed)
    if not config.style_target and (config.w_style > 0 or not config.w_content):
        # demo mode: a style term without a style image -> seeded synthetic style image AND (explicitly) synthetic
        # loss-network filters; a real run needs --style_target and data/model/vgg_19.npz or
        # tensorflow_inception_graph.npz (the loaders raise without them)
        from neural_flow_style_amd import synthetic as S
        print("DEMO MODE: synthetic style image and synthetic (random) loss-network filters -- not a real stylisation")
        config.style_target = S.style_image(256, 256, np.random.RandomState(config.seed))
        config.w_style = 1
        config.synthetic_weights = True
    styler = Styler(config)
    print("loss network weights:", styler.net.source)
    styler.load_img(config.resolution[1:])
    params = load_frames(config) or synthetic_frames(config)
    result = styler.run(params)

    from PIL import Image
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        for o, l_ in enumerate(result["l"]):
            plt.plot(range(len(l_)), l_, label="oct %d" % o)
        plt.legend()
        plt.savefig(os.path.join(config.log_dir, "loss_plot.png"))
    except Exception as e:  # plotting is optional
        print("loss plot skipped:", e)
    for i, img in enumerate(result["r"]):
        Image.fromarray(img).save(os.path.join(config.log_dir, "%03d.png" % (config.target_frame + i)))
    for i, d in enumerate(result["d"]):
        np.savez_compressed(os.path.join(config.log_dir, "%03d.npz" % (config.target_frame + i)), x=d[:, ::-1])
    for o, d_intm_o in enumerate(result["d_intm"]):          # intermediate octave images (test_smokegun.py:103-109)
        for i, img in enumerate(d_intm_o):
            if img is None:
                continue
            Image.fromarray(img).save(os.path.join(config.log_dir, "o%02d_%03d.png" % (o, config.target_frame + i)))
    return result


def main(config):
    """The reference's main() (test_smokegun.py:111-197), value for value: dataset, num_kernels 2, kernel_scale 2,
    support 4, disc 1, radius 0.5, nsize 1, rest_density 1000, resolution / domain [200,300,200], clip False, w_density
    0, k 3, window_sigma 3, batch_size 1, frames_per_opt 1, target_field 'd', lr 0.1, network
    'tensorflow_inception_graph.pb' with style layers ['conv2d2','mixed3b','mixed4b'] x [1,1,1], octave_n 1,
    octave_scale 1.8, transmit 0.01, iter 20, resize_scale 300/resolution[0], rotate False, interp 1.

    The reference sets them unconditionally; here a flag given on the command line wins over the block (SURVEY.md
    section 0.1: the block must let BASELINE's configuration through), for these flags only: ``--network``,
    ``--style_layer`` / ``--w_style_layer``, ``--rotate`` / ``--n_views``, ``--resolution``, ``--iter``, ``--transmit``,
    ``--d_path``.  BASELINE configs[2] is

        python test_smokegun.py --network vgg_19.ckpt --rotate true --n_views 8 --w_style 1 --w_content 0 ...

    (with ``--network vgg_19.ckpt`` and no ``--style_layer`` the style layers are conv1_1 ... conv5_1, and a content
    layer that is not a VGG end point switches the content term off).  d_path defaults to 'pt_low_o2/%03d.npz'
    (``io_bgeo`` reads the reference's '.bgeo' when that file exists instead)."""
    import sys
    given = set(a.split("=")[0] for a in sys.argv[1:] if a.startswith("--"))
    config.dataset = "smokegun"
    if "--d_path" not in given:
        config.d_path = "pt_low_o2/%03d.npz"
    config.num_kernels = 2
    config.kernel_scale = 2
    config.support = 4
    config.disc = 1
    config.radius = 1 / config.disc / 2
    config.nsize = 1
    config.rest_density = 1000
    if "--resolution" not in given:
        config.resolution = [200, 300, 200]
    config.domain = list(config.resolution)
    config.clip = False
    config.w_density = 0
    config.k = 3
    config.window_sigma = 3
    config.batch_size = 1
    config.frames_per_opt = 1
    config.target_field = "d"
    config.lr = 0.1
    if "--network" not in given:
        config.network = "tensorflow_inception_graph.pb"
    if "--style_layer" not in given:
        if "vgg" in config.network:
            config.style_layer = ["conv1_1", "conv2_1", "conv3_1", "conv4_1", "conv5_1"]
        else:
            config.style_layer = ["conv2d2", "mixed3b", "mixed4b"]
    if "--w_style_layer" not in given:
        config.w_style_layer = [1] * len(config.style_layer)
    if "vgg" in config.network and not str(config.content_layer).startswith("conv"):
        config.w_content = 0          # the default content layer is an Inception-v1 tensor: style transfer only
    config.octave_n = 1
    config.octave_scale = 1.8
    if "--transmit" not in given:
        config.transmit = 0.01
    if "--iter" not in given:
        config.iter = 20
    config.resize_scale = 300 / config.resolution[0]
    if "--rotate" not in given:
        config.rotate = False
    config.interp = 1
    return run(config)


if __name__ == "__main__":
    cfg, _ = get_config()
    main(cfg)
