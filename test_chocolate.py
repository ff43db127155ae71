"""Counterpart of the reference driver ``test_chocolate.py`` (run: test_chocolate.py:14-146, main: 148-252) on the
MI355X build -- BASELINE ``configs[4]``: the SPH "chocolate" liquid, Lagrangian stylisation of the particle POSITIONS
('p' field: the variable is a displacement per particle) through the density splat, the liquid render
(``1 - exp(-tau sum d)``, ``transmit`` 0.2), two octaves, optional pressure term.

Same ``run(config)`` body as the reference: build the Styler, ``load_img(resolution[1:])``, read the particle frames
(``id, position``; slots of absent particles stay at -1, i.e. outside the domain), ``styler.run(params)``, save
``loss_plot.png``, ``%03d.bgeo`` (``position`` de-normalised back to (x,y,z) world units + ``radius``; padded slots
skipped), ``%03d.png`` (``result['r']``), ``o%02d_%03d.png``.  ``main()`` reproduces the reference's override block
value for value (``tensorflow_inception_graph.pb`` with style layers 'conv2d2', 'mixed3b', 'mixed4b';
``--network vgg_19.ckpt`` selects VGG-19 conv1_1 ... conv4_1).  Differences, each because the reference's choice cannot
run here:
  * particle files go through ``io_bgeo`` or ``.npz`` (keys ``position`` [N,3] world (x,y,z), optional ``id``); without
    a dataset the run uses seeded synthetic particles (DEMO MODE, printed);
  * ``np.float`` (test_chocolate.py:121, removed from NumPy) is ``float``; no open3d viewer at the end.

    python test_chocolate.py --style_target data/image/pattern1.png --target_frame 90 --num_frames 1
    python -m torch.distributed.run --nproc-per-node 8 test_chocolate.py --num_frames 120 ...   (frames over 8 GPUs)
"""
import os

import numpy as np

from config import get_config
from styler_3p import Styler
from util import prepare_dirs_and_logger


def load_frames(config):
    """test_chocolate.py:27-66: nmax over the frames, p_[id_j] = position[id_j], padded slots at -1; normalised by the
    domain and ordered (z,y,x)"""
    raw, nmax = [], 0
    for i in range(config.num_frames):
        path = os.path.join(config.data_dir, config.dataset, config.d_path % (config.target_frame + i))
        npz = os.path.splitext(path)[0] + ".npz"
        if os.path.exists(path) and path.endswith(".bgeo"):
            import io_bgeo as partio
            pt = partio.read(path)
            ids = pt.array("id")[:, 0]
            raw.append((ids, pt.array("position")))
        elif os.path.exists(npz):
            z = np.load(npz)
            pos = np.asarray(z["position"], np.float32)
            ids = np.asarray(z["id"]).reshape(-1) if "id" in z.files else np.arange(pos.shape[0])
            raw.append((ids, pos))
        elif i == 0:
            return None                                        # no dataset at all: the caller's demo mode
        else:
            raise FileNotFoundError("frame %d of the sequence is missing: %s (frame 0 exists -- refusing to replace a "
                                    "partly present dataset by synthetic particles)" % (i, path))
        nmax = max(nmax, int(raw[-1][0].max()) + 1, raw[-1][1].shape[0])
    p = []
    for ids, pos in raw:
        p_ = np.ones([nmax, 3], np.float32) * -1
        p_[ids] = pos[ids]
        px, py, pz = p_[:, 0] / config.domain[2], p_[:, 1] / config.domain[1], p_[:, 2] / config.domain[0]
        p.append(np.stack([pz, py, px], -1).astype(np.float32))
    return {"p": p}


def synthetic_frames(config, n=50000):
    """seeded liquid-like blobs; the same particles in every frame, drifting (the temporal filter needs particle i to be
    the same particle in every frame)"""
    from neural_flow_style_amd import synthetic as S
    rng = np.random.RandomState(config.seed)
    p0 = S.blob_particles(n, rng)
    drift = rng.randn(n, 3).astype(np.float32) * 0.002
    return {"p": [np.clip(p0 + drift * t, 0.02, 0.98).astype(np.float32) for t in range(config.num_frames)]}


def run(config):
    prepare_dirs_and_logger(config)
    config.rng = np.random.RandomState(config.seed)
    if not config.style_target and (config.w_style > 0 or not config.w_content):
        from neural_flow_style_amd import synthetic as S
        print("DEMO MODE: synthetic style image and synthetic (random) loss-network filters -- not a real stylisation")
        config.style_target = S.style_image(256, 256, np.random.RandomState(config.seed))
        config.w_style = 1
        config.synthetic_weights = True

    styler = Styler(config)
    print("loss network weights:", styler.net.source)
    # frames over the ranks of a launcher (one process per GPU; SURVEY 8(e) "Frames (config 4/5)")
    import torch.distributed as dist
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
            dist.init_process_group(os.environ.get("NFS_DIST_BACKEND", "nccl"))
        styler.pg = dist.group.WORLD
        styler.shard_by = "frames" if config.num_frames > 1 else "views"
        if styler.shard_by == "frames":
            # one Adam state per optimiser group (styler_3p.py:315-323) is updated frame after frame, so a group cannot
            # straddle ranks: a sharded run caps the group at a rank's block of frames ("in sharded mode
            # give each frame group its own state") -- with the driver's 120 the whole sequence would sit on one rank
            world = dist.get_world_size()
            per_rank = -(-config.num_frames // (world * max(config.interp, 1))) * max(config.interp, 1)
            if styler.frames_per_opt > per_rank:
                print("frames_per_opt %d -> %d (one optimiser group per rank's block of frames)"
                      % (styler.frames_per_opt, per_rank))
                styler.frames_per_opt = per_rank
    styler.load_img(config.resolution[1:])

    params = load_frames(config)
    if params is None:
        print("DEMO MODE: no particle files under %s -- seeded synthetic particles"
              % os.path.join(config.data_dir, config.dataset))
        params = synthetic_frames(config)
    print("resolution:", config.resolution)
    print("domain:", config.domain)
    print("radius:", config.radius)

    result = styler.run(params)
    if int(os.environ.get("RANK", "0")) != 0:
        return result

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

    # stylised particles (test_chocolate.py:91-125): de-normalise to (x,y,z) world units, skip the padded slots
    import io_bgeo as partio
    for i, p_sty in enumerate(result["p"]):
        px, py, pz = p_sty[..., 2] * config.domain[2], p_sty[..., 1] * config.domain[1], p_sty[..., 0] * config.domain[0]
        xyz = np.stack([px, py, pz], -1)
        keep = xyz[:, 0] >= 0
        pt = partio.create()
        pt.addAttribute("position", partio.VECTOR, 3)
        pt.addAttribute("radius", partio.FLOAT, 1)
        pt.addParticles(int(keep.sum()))
        pt.set_array("position", xyz[keep])
        pt.set_array("radius", np.full((int(keep.sum()), 1), config.radius, np.float32))
        partio.write(os.path.join(config.log_dir, "%03d.bgeo" % (config.target_frame + i)), pt)
    for i, r_sty_ in enumerate(result["r"]):
        Image.fromarray(r_sty_).save(os.path.join(config.log_dir, "%03d.png" % (config.target_frame + i)))
    for o, d_intm_o in enumerate(result["d_intm"]):
        for i, d_intm_ in enumerate(d_intm_o):
            if d_intm_ is None:
                continue
            Image.fromarray(d_intm_).save(os.path.join(config.log_dir, "o%02d_%03d.png" % (o, config.target_frame + i)))
    return result


def main(config):
    """The reference's main() (test_chocolate.py:148-252), value for value; ``--iter`` / ``--octave_n`` /
    ``--resolution`` / ``--w_pressure`` given on the command line win over the block's values."""
    import sys
    given = set(a.split("=")[0] for a in sys.argv[1:] if a.startswith("--"))
    config.dataset = "chocolate"
    config.d_path = "partio/ParticleData_Fluid_%d.bgeo"

    # from scene
    config.radius = 0.025
    config.support = 4
    config.disc = 2
    config.rest_density = 1000
    sim_resolution = [128, 128, 128]                                # original resolution [D,H,W]
    cell_size = 2 * config.radius * config.disc
    config.domain = [float(_ * cell_size) for _ in sim_resolution]
    config.nsize = max(3 - config.disc, 1)

    # upscaling for rendering
    if "--resolution" not in given:
        config.resolution = [200, 200, 200]

    # default settings
    config.lr = 0.002
    if "--iter" not in given:
        config.iter = 20
    config.resize_scale = 1
    config.transmit = 0.2
    config.clip = False
    config.num_kernels = 1
    config.k = 3
    if "--octave_n" not in given:
        config.octave_n = 2
    config.octave_scale = 1.8
    config.render_liquid = True
    config.rotate = False

    # test_chocolate.py:173-180: the Inception graph with conv2d2 / mixed3b / mixed4b; ``--network vgg_19.ckpt`` selects
    # the VGG-19 end points conv1_1 ... conv4_1 instead
    if "--network" not in given:
        config.network = "tensorflow_inception_graph.pb"
    if "--style_layer" not in given:
        if "vgg" in config.network:
            config.style_layer = ["conv1_1", "conv2_1", "conv3_1", "conv4_1"]
        else:
            config.style_layer = ["conv2d2", "mixed3b", "mixed4b"]
    if "--w_style_layer" not in given:
        config.w_style_layer = [1] * len(config.style_layer)
    if "vgg" in config.network and not str(config.content_layer).startswith("conv"):
        config.w_content = 0

    # frame range setting
    config.frames_per_opt = 120
    config.batch_size = 1
    config.window_sigma = 9

    # position test
    config.target_field = "p"
    style = os.path.splitext(os.path.basename(config.style_target))[0] if config.style_target else "synthetic"
    config.tag = "test_%s_%s_%d_intp%d" % (config.target_field, style, config.num_frames, config.interp)
    return run(config)


if __name__ == "__main__":
    config, unparsed = get_config()
    main(config)
