"""Counterpart of the reference driver ``test_smokegun_resim.py`` (test_smokegun_resim.py:219-387) on the
MI355X build: per frame load the density (npz key ``x``, H flipped on read) and the mantaflow MAC velocity,
convert to cell-centred normalised (z,y,x) velocity, run the SimG2P resampler (or naive advection), and save
the particle set -- as ``%03d.npz`` with keys ``id``, ``position`` (world units, (x,y,z), y up), ``density``
[N,octave_n], ``radius`` (the attributes the reference writes to partio ``.bgeo``; this is the format
``test_smokegun.py`` of this repo reads) -- plus the density preview ``%03d.png`` and ``stat.txt``.
Without a dataset it runs on a seeded synthetic plume.

    python test_smokegun_resim.py --num_frames 3
"""
import os

import numpy as np

from config import get_config
from util import prepare_dirs_and_logger
from neural_flow_style_amd.resim import SimG2P, mac_to_centered, velocity_to_normalised


def load_frame(config, t):
    d_path = os.path.join(config.data_dir, config.dataset, config.d_path % (config.target_frame + t))
    v_path = os.path.join(config.data_dir, config.dataset, config.v_path % (config.target_frame + t))
    if not (os.path.exists(d_path) and os.path.exists(v_path)):
        return None
    with np.load(d_path) as data:
        d = data["x"][:, ::-1]                                   # [D,H,W], [0-1]
    with np.load(v_path) as data:
        v_ = mac_to_centered(data["x"])                          # [D,H,W,3] (x,y,z), H flipped
    return np.ascontiguousarray(d, np.float32), velocity_to_normalised(v_, config.scale)


def synthetic_frame(config, t):
    """a rising Gaussian puff with a swirling velocity (cells/frame, MAC layout), seeded"""
    D, H, W = config.resolution
    rng = np.random.RandomState(config.seed + t)
    zz, yy, xx = np.meshgrid(np.linspace(0, 1, D), np.linspace(0, 1, H), np.linspace(0, 1, W), indexing="ij")
    cy = 0.75 - 0.03 * t
    d = np.exp(-((zz - 0.5) ** 2 + (yy - cy) ** 2 + (xx - 0.5) ** 2) / 0.01).astype(np.float32)
    d[d < 0.02] = 0
    mac = np.zeros((D, H, W, 3), np.float32)
    mac[..., 1] = 1.5                                            # upwards (y up in simulation space)
    mac[..., 0] = 0.5 * np.sin(2 * np.pi * zz)
    mac[..., 2] = 0.5 * np.cos(2 * np.pi * xx)
    mac += rng.randn(D, H, W, 3).astype(np.float32) * 0.05
    return d[:, ::-1].copy(), velocity_to_normalised(mac_to_centered(mac), config.scale)


def run(config):
    prepare_dirs_and_logger(config)
    config.rng = np.random.RandomState(config.seed)
    resampler = SimG2P(config)
    p = p_id = p_src = None
    n_prev, l = 0, 0.0
    for t in range(config.num_frames):
        frame = load_frame(config, t) or synthetic_frame(config, t)
        d, u = frame
        if config.resampling:
            if t == 0:
                p, p_id = resampler.sample(d, disc=config.disc, threshold=0)     # initial seeding, no optimisation
            result = resampler.optimize(p, p_id, d, u)
            p, p_id, p_den = result["p"], result["p_id"], result["p_den"]
            l = result["l"][-1] if result["l"] else 0.0
            d_smp = result["d_smp"]
        else:
            if t == 0:
                p, p_id = resampler.sample(d, disc=config.disc, threshold=0)
                p_src = p
            else:
                p = np.concatenate([p, p_src], axis=0)           # re-emit the source particles of t = 0
                p_id = np.arange(p.shape[0])
            p_den = np.ones([p.shape[0], 1], np.float32)
            p, d_smp = resampler.naive_adv(p, u, p_den)
            l = 0.0
        print(t, "num particles", p.shape[0], "(+%d)" % (p.shape[0] - n_prev), "loss", l)
        n_prev = p.shape[0]

        # back to the original domain coordinates (x,y,z), y up  (test_smokegun_resim.py:288-293)
        p_ = np.stack([p[..., 2] * config.domain[2], (1 - p[..., 1]) * config.domain[1], p[..., 0] * config.domain[0]],
                      axis=-1).astype(np.float32)
        np.savez_compressed(os.path.join(config.log_dir, "%03d.npz" % (config.target_frame + t)),
                            id=np.asarray(p_id, np.int64), position=p_, density=np.asarray(p_den, np.float32),
                            radius=np.float32(config.radius))
        # ... and as the reference writes it (test_smokegun_resim.py:295-319): partio .bgeo with id, position, density,
        # Cd (the first density channel three times) and radius
        import io_bgeo as partio
        pt = partio.from_arrays({"id": np.asarray(p_id, np.int32), "position": p_,
                                 "density": np.asarray(p_den, np.float32),
                                 "Cd": np.repeat(np.asarray(p_den, np.float32)[:, :1], 3, axis=1),
                                 "radius": np.full((p_.shape[0], 1), config.radius, np.float32)},
                                types={"density": partio.VECTOR if np.asarray(p_den).shape[1] > 1 else partio.FLOAT})
        partio.write(os.path.join(config.log_dir, "%03d.bgeo" % (config.target_frame + t)), pt)
        # density preview, same transmittance render as the reference (326-331)
        transmit = np.exp(-np.cumsum(d_smp[::-1], axis=0) * config.transmit)
        d_img = np.sum(d_smp * transmit, axis=0)
        d_img /= max(float(d_img.max()), 1e-12)
        try:
            from PIL import Image
            Image.fromarray((d_img[::-1] * 255).astype(np.uint8)).save(
                os.path.join(config.log_dir, "%03d.png" % (config.target_frame + t)))
        except ImportError:
            pass
    with open(os.path.join(config.log_dir, "stat.txt"), "w") as f:
        f.write("num particles %d\n" % p.shape[0])
        f.write("loss %.2f" % l)
    return p, p_id


def main(config):
    config.dataset = "smokegun"
    config.d_path = "d_low/%03d.npz"
    config.v_path = "v_low/%03d.npz"
    config.target_frame = 0
    config.scale = 1
    have_data = os.path.exists(os.path.join(config.data_dir, config.dataset, config.d_path % 0))
    config.domain = [_ * config.scale for _ in ([200, 300, 200] if have_data else [48, 64, 48])]
    config.resolution = [int(_) for _ in config.domain]
    config.disc = 1
    config.radius = 1 / config.disc / 2                          # cell_size = 2 * radius * disc = 1
    config.nsize = 1
    config.support = 4
    config.rest_density = 1000
    config.threshold = 0.01
    config.lr = 0.0005
    config.iter = 20
    config.transmit = 0.01
    config.octave_n = 2
    config.octave_scale = 2 if config.octave_n > 1 else 1
    config.resampling = True
    config.tag = ("n%d_it%d_o%d" % (config.num_frames, config.iter, config.octave_n)) if config.resampling \
        else "naive_n%d" % config.num_frames
    run(config)


if __name__ == "__main__":
    config, unparsed = get_config()
    main(config)
