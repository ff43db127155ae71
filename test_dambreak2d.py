"""Counterpart of the reference driver ``test_dambreak2d.py`` (run: test_dambreak2d.py:14-131, main: 133-192) on the
MI355X build -- BASELINE ``configs[0]``: the 2-D SPH dam break, per-particle COLOUR stylised through the colour splat,
VGG-19 ``conv2_1`` / ``conv3_1`` Gram style loss masked by the density (``style_mask``), TV term, three octaves.

Same ``run(config)`` body as the reference: build the Styler, ``load_img(resolution)``, read the particle frames
(``id, position, density``), ``styler.run(params)``, save ``loss_plot.png``, ``%03d.png`` (``result['d']``),
``o%02d_%03d.png`` (``result['d_intm']``) and ``%03d.bgeo`` with ``position`` (2 components), ``Cd``, ``radius``.
``main()`` reproduces the reference's override block value for value.  Differences, each because the reference's
choice cannot run here:
  * particle files go through ``io_bgeo`` (classic .bgeo, the calls the reference makes on partio) or ``.npz``
    (keys ``position`` [N,2|3] world units (x,y[,z]), ``density`` [N,1]); without a dataset the run uses a seeded
    synthetic dam-break block (DEMO MODE, printed);
  * without ``--style_target`` a seeded synthetic style image and -- explicitly -- synthetic VGG filters are used (a real
    run needs data/model/vgg_19.npz: ``vgg.load_vgg`` raises without it);
  * no open3d viewer at the end.

    python test_dambreak2d.py --style_target data/image/wave.jpeg --target_frame 150 --num_frames 1
"""
import os

import numpy as np

from config import get_config
from styler_2p import Styler
from util import prepare_dirs_and_logger


def load_frames(config):
    """test_dambreak2d.py:27-55: particle j reads the position / density stored at index id[j]; positions normalised by
    the domain and ordered (y,x)"""
    p, r = [], []
    for i in range(config.num_frames):
        path = os.path.join(config.data_dir, config.dataset, config.d_path % (config.target_frame + i))
        npz = os.path.splitext(path)[0] + ".npz"
        if os.path.exists(path) and path.endswith(".bgeo"):
            import io_bgeo as partio
            pt = partio.read(path)
            ids = pt.array("id")[:, 0]
            pos = np.zeros([pt.numParticles(), 2], np.float32)
            den = np.zeros([pt.numParticles(), 1], np.float32)
            pos[ids] = pt.array("position")[ids][:, :2]            # 2d: drop z (line 44)
            den[ids] = pt.array("density")[ids][:, :1]
        elif os.path.exists(npz):
            z = np.load(npz)
            pos = np.asarray(z["position"], np.float32)[:, :2]
            den = np.asarray(z["density"], np.float32).reshape(pos.shape[0], -1)[:, :1]
        elif i == 0:
            return None                                        # no dataset at all: the caller's demo mode
        else:
            raise FileNotFoundError("frame %d of the sequence is missing: %s (frame 0 exists -- refusing to replace a "
                                    "partly present dataset by synthetic particles)" % (i, path))
        px, py = pos[:, 0] / config.domain[1], pos[:, 1] / config.domain[0]
        p.append(np.stack([py, px], -1).astype(np.float32))
        r.append(den)
    return {"p": p, "r": r}


def synthetic_frames(config):
    """a seeded dam-break block: jittered lattice in the lower-left 35 % x 60 % of the unit square, rest density +-10 %"""
    from neural_flow_style_amd import synthetic as S
    rng = np.random.RandomState(config.seed)
    # one particle per simulation cell of the UN-scaled resolution (resolution / scale), like the scene
    n_side = max(int(config.resolution[0] / max(getattr(config, "scale", 1), 1)), 16)
    p0 = S.dambreak_particles(n_side, rng)
    p0[:, 0] = 1.0 - p0[:, 0]                       # y up in simulation space: the block sits on the floor
    p, r = [], []
    for t in range(config.num_frames):
        shift = np.array([0.0, 0.004 * t], np.float32)              # drifts right frame by frame
        p.append(np.clip(p0 + shift, 0.0, 0.999).astype(np.float32))
        r.append(rng.uniform(0.9, 1.1, (p0.shape[0], 1)).astype(np.float32) * config.rest_density)
    return {"p": p, "r": r}


def run(config):
    prepare_dirs_and_logger(config)
    config.rng = np.random.RandomState(config.seed)
    if not config.style_target:
        from neural_flow_style_amd import synthetic as S
        print("DEMO MODE: synthetic style image and synthetic (random) VGG-19 filters -- not a stylisation by VGG-19")
        config.style_target = S.style_image(256, 256, np.random.RandomState(config.seed))
        config.synthetic_weights = True
    styler = Styler(config)
    print("loss network weights:", styler.net.source)
    styler.load_img(config.resolution)

    params = load_frames(config)
    if params is None:
        print("DEMO MODE: no particle files under %s -- seeded synthetic dam-break particles"
              % os.path.join(config.data_dir, config.dataset))
        params = synthetic_frames(config)
    p = params["p"]
    print("resolution:", config.resolution)
    print("domain:", config.domain)
    print("radius:", config.radius)
    print("num particles:", p[0].shape)

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

    for i, d_sty_ in enumerate(result["d"]):                       # uint8 [H,W,3]
        Image.fromarray(d_sty_).save(os.path.join(config.log_dir, "%03d.png" % (config.target_frame + i)))
    for o, d_intm_o in enumerate(result["d_intm"]):
        for i, d_intm_ in enumerate(d_intm_o):
            # (the reference names every image of an octave o%02d_<target_frame>.png, line 96: later frames overwrite
            # earlier ones; here the frame index is kept)
            Image.fromarray(d_intm_).save(os.path.join(config.log_dir, "o%02d_%03d.png" % (o, config.target_frame + i)))

    # particles with their colours (test_dambreak2d.py:99-124): positions de-normalised back to (x,y) world units
    import io_bgeo as partio
    c_sty = result["c"]
    for i in range(config.num_frames):
        px, py = p[i][..., 1] * config.domain[1], p[i][..., 0] * config.domain[0]
        pt = partio.create()
        position = pt.addAttribute("position", partio.VECTOR, 2)
        color = pt.addAttribute("Cd", partio.FLOAT, 3)
        radius = pt.addAttribute("radius", partio.FLOAT, 1)
        first = pt.addParticles(p[i].shape[0])
        assert first == 0
        pt.set_array("position", np.stack([px, py], -1))
        pt.set_array("Cd", c_sty[i])
        pt.set_array("radius", np.full((p[i].shape[0], 1), config.radius, np.float32))
        partio.write(os.path.join(config.log_dir, "%03d.bgeo" % (config.target_frame + i)), pt)
        del position, color, radius
    return result


def main(config):
    """The reference's main() (test_dambreak2d.py:133-192), value for value.  ``--iter`` / ``--octave_n`` /
    ``--num_frames`` given on the command line win over the block's values (so that a short run can be asked for); the
    default style layers / weights are applied when the flags are left at their defaults."""
    import sys
    given = set(a.split("=")[0] for a in sys.argv[1:] if a.startswith("--"))
    config.dataset = "dambreak2d"
    config.d_path = "partio/ParticleData_Fluid_%d.bgeo"

    # from scene
    config.radius = 0.025
    config.support = 4
    config.disc = 2
    config.rest_density = 1000
    config.resolution = [128, 256]                                  # [H,W]
    cell_size = 2 * config.radius * config.disc
    config.domain = [float(_ * cell_size) for _ in config.resolution]
    config.nsize = max(3 - config.disc, 1)

    # upscaling for rendering
    if "--scale" not in given:
        config.scale = 4
    config.scale = int(config.scale)
    config.nsize *= config.scale
    config.resolution = [config.resolution[0] * config.scale, config.resolution[1] * config.scale]

    config.frames_per_opt = 200
    config.window_sigma = 3

    # colour test
    config.target_field = "c"
    config.lr = 0.01
    if "--iter" not in given:
        config.iter = 100
    if "--octave_n" not in given:
        config.octave_n = 3
    config.octave_scale = 1.7
    config.clip = False

    config.network = "vgg_19.ckpt"
    config.w_style = 1
    config.w_content = 0
    config.style_init = "noise"
    if "--style_layer" not in given:
        config.style_layer = ["conv2_1", "conv3_1"]
        config.w_style_layer = [0.5, 0.5]
    config.style_mask = True
    config.style_mask_on_ref = False
    config.style_tiling = 2
    config.w_tv = 0.01
    style = os.path.splitext(os.path.basename(config.style_target))[0] if config.style_target else "synthetic"
    config.tag = "test_%s_%s" % (config.target_field, style)
    return run(config)


if __name__ == "__main__":
    config, unparsed = get_config()
    main(config)
