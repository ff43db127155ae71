"""Flag surface of the reference's ``config.py`` (config.py:15-114): same flag names, types and
defaults, so a reference driver's ``config`` Namespace drops in.  ``get_config()`` returns
``(Namespace, unparsed)``.  Drivers add non-flag attributes afterwards (``rng``, ``num_kernels``,
``kernel_scale``; test_smokegun.py:22,117-120) -- ``complete()`` fills their defaults.

Build-specific additions (not in the reference): ``--views_mode`` (``sequential`` = the reference's
one-Adam-step-per-view loop, ``sum`` = gradient of the summed view losses, shardable over GPUs),
``--grid_variable`` / ``--transport_recursive`` for the grid (TNST-style) path, ``--synthetic_weights`` (explicit
opt-in to seeded synthetic VGG filters when no converted checkpoint exists), ``--ray_mode`` (``max`` / ``mean``: the
per-ray reductions north_star names beside the reference's transmittance / liquid integrals; '' follows
``render_liquid``).  The reference's own ``--optimizer`` flag (config.py:78, default 'adam', the only value its code
honours) additionally takes ``lbfgs`` here: limited-memory BFGS, north_star's other outer loop (engine.LBFGSState).
"""
from __future__ import annotations

import argparse

import numpy as np


def str2bool(v):
    return str(v).lower() in ("true", "1", "yes", "y", "t")


# (group, flag, kwargs) -- one row per reference flag
_FLAGS = [
    ("Path", "data_dir", dict(type=str, default="data")),
    ("Path", "log_dir", dict(type=str, default="log")),
    ("Path", "model_dir", dict(type=str, default="model")),
    ("Path", "d_path", dict(type=str, default="d/%03d.npz")),
    ("Path", "v_path", dict(type=str, default="v/%03d.npz")),
    ("Path", "tag", dict(type=str, default="test")),
    ("Data", "dataset", dict(type=str, default="smokegun")),
    ("Data", "target_frame", dict(type=int, default=70)),
    ("Data", "num_frames", dict(type=int, default=1)),
    ("Data", "scale", dict(type=float, default=2.0)),
    ("Network", "network", dict(type=str, default="tensorflow_inception_graph.pb",
                                choices=["tensorflow_inception_graph.pb", "vgg_19.ckpt"])),
    ("Network", "pool1", dict(type=str2bool, default=False)),
    ("Network", "batch_size", dict(type=int, default=1)),
    ("Grid", "resolution", dict(nargs="+", type=int, default=[384, 288])),
    ("Grid", "adv_order", dict(type=int, default=1, choices=[1, 2])),
    ("Particle", "domain", dict(nargs="+", type=int, default=[12.8, 12.8, 12.8])),
    ("Particle", "radius", dict(type=float, default=0.025)),
    ("Particle", "disc", dict(type=int, default=2)),
    ("Particle", "nsize", dict(type=int, default=1)),
    ("Particle", "rest_density", dict(type=float, default=1000)),
    ("Particle", "w_pressure", dict(type=float, default=0)),
    ("Particle", "w_density", dict(type=float, default=0)),
    ("Particle", "window_sigma", dict(type=float, default=2)),
    ("Particle", "interp", dict(type=int, default=1)),
    ("Particle", "support", dict(type=float, default=4)),
    ("Particle", "k", dict(type=int, default=3)),
    ("Particle", "clip", dict(type=str2bool, default=False)),
    ("Render", "resize_scale", dict(type=float, default=1.0)),
    ("Render", "transmit", dict(type=float, default=0.01)),
    ("Render", "rotate", dict(type=str2bool, default=False)),
    ("Render", "phi0", dict(type=int, default=-5)),
    ("Render", "phi1", dict(type=int, default=5)),
    ("Render", "phi_unit", dict(type=int, default=5)),
    ("Render", "theta0", dict(type=int, default=-10)),
    ("Render", "theta1", dict(type=int, default=10)),
    ("Render", "theta_unit", dict(type=int, default=10)),
    ("Render", "v_batch", dict(type=int, default=1)),
    ("Render", "n_views", dict(type=int, default=9)),
    ("Render", "sample_type", dict(type=str, default="poisson", choices=["uniform", "poisson", "both"])),
    ("Render", "render_liquid", dict(type=str2bool, default=False)),
    ("Optimizer", "target_field", dict(type=str, default="p", choices=["d", "p", "c"])),
    ("Optimizer", "optimizer", dict(type=str, default="adam")),
    ("Optimizer", "iter", dict(type=int, default=20)),
    ("Optimizer", "lr", dict(type=float, default=0.0007)),
    ("Optimizer", "lr_scale", dict(type=float, default=1)),
    ("Optimizer", "octave_n", dict(type=int, default=2)),
    ("Optimizer", "octave_scale", dict(type=float, default=1.8)),
    ("Optimizer", "frames_per_opt", dict(type=int, default=10)),
    ("Style", "content_layer", dict(type=str, default="mixed4d_3x3_bottleneck_pre_relu")),
    ("Style", "content_channel", dict(type=int, default=139)),
    ("Style", "w_content", dict(type=float, default=1)),
    ("Style", "w_content_amp", dict(type=float, default=100)),
    ("Style", "content_target", dict(type=str, default="")),
    ("Style", "top_k", dict(type=int, default=5)),
    ("Style", "style_layer", dict(nargs="+", type=str, default=["conv3_1"])),
    ("Style", "w_style", dict(type=float, default=0)),
    ("Style", "w_style_layer", dict(nargs="+", type=float, default=[1])),
    ("Style", "hist_layer", dict(nargs="+", type=str, default=["input"])),
    ("Style", "w_hist", dict(type=float, default=0)),
    ("Style", "w_hist_layer", dict(nargs="+", type=float, default=[1])),
    ("Style", "w_tv", dict(type=float, default=0)),
    ("Style", "style_target", dict(type=str, default="")),
    ("Style", "style_mask", dict(type=str2bool, default=False)),
    ("Style", "style_mask_on_ref", dict(type=str2bool, default=False)),
    ("Style", "style_tiling", dict(type=int, default=1)),
    ("Style", "style_init", dict(type=str, default="noise", choices=["noise", "style"])),
    ("Misc", "seed", dict(type=int, default=123)),
    ("Misc", "gpu_id", dict(type=str, default="0")),
    # ---- build-specific -------------------------------------------------------------------
    ("MI355X", "views_mode", dict(type=str, default="sequential", choices=["sequential", "sum"])),
    ("MI355X", "grid_variable", dict(type=str, default="", choices=["", "v", "d"])),
    ("MI355X", "synthetic_weights", dict(type=str2bool, default=False)),
    ("MI355X", "transport_recursive", dict(type=str2bool, default=True)),
    ("MI355X", "ray_mode", dict(type=str, default="", choices=["", "transmit", "liquid", "max", "mean"])),
]


def make_parser():
    parser = argparse.ArgumentParser()
    groups = {}
    for grp, flag, kw in _FLAGS:
        g = groups.get(grp)
        if g is None:
            g = groups[grp] = parser.add_argument_group(grp)
        g.add_argument("--" + flag, **kw)
    return parser


parser = make_parser()


def get_config(argv=None):
    config, unparsed = parser.parse_known_args(argv)
    return config, unparsed


def complete(config):
    """defaults for the attributes the reference drivers inject after parsing"""
    if not hasattr(config, "rng") or config.rng is None:
        config.rng = np.random.RandomState(config.seed)
    if not hasattr(config, "num_kernels"):
        config.num_kernels = 1
    if not hasattr(config, "kernel_scale"):
        config.kernel_scale = 2
    if not hasattr(config, "views_mode"):
        config.views_mode = "sequential"
    if not hasattr(config, "grid_variable"):
        config.grid_variable = ""
    if not hasattr(config, "synthetic_weights"):
        config.synthetic_weights = False
    if not hasattr(config, "transport_recursive"):
        config.transport_recursive = True
    if not hasattr(config, "ray_mode"):
        config.ray_mode = ""

    return config
