#!/usr/bin/env python
"""Golden vectors for the host helpers of the stylisation loop from the REFERENCE ITSELF: its util.py is imported in
the build container and ``denoise`` (temporal smoothing of the per-frame updates, styler_3p.py:377-381) and
``crop_ratio`` (style / content image crop, styler_base.py:320-338) are run on seeded inputs; the Laplacian-pyramid
kernels ``k5x5`` / ``k5x5x5`` (util.py:27-46) and ``cosine_decay`` values are dumped as the module computes them, and
the small NumPy helpers of the drivers' surface (rgb2yuv / yuv2rgb, make_grid, str2bool) are run on seeded inputs.  util.py imports
imageio, skimage, tensorflow, matplotlib and open3d at module level; modules of those names that are absent here are
registered empty only to let the import statements pass -- neither function touches them (``resize`` does, through
skimage, and therefore is not part of this fixture).   Run:  python tests/golden/make_util_fixture.py
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np


def _stub(name, attrs=()):
    try:
        importlib.import_module(name)
        return
    except Exception:
        pass
    m = types.ModuleType(name)
    for a in attrs:
        setattr(m, a, types.ModuleType(name + "." + a))
        sys.modules[name + "." + a] = getattr(m, a)
    sys.modules[name] = m


_stub("imageio")
_stub("skimage", ("transform",))
_stub("matplotlib", ("colors", "pyplot", "cm"))
_stub("open3d")
tf = types.ModuleType("tensorflow")
tf.image = types.SimpleNamespace(ResizeMethod=types.SimpleNamespace(NEAREST_NEIGHBOR=0, BILINEAR=1, BICUBIC=2))
tf.float32 = "float32"
sys.modules["tensorflow"] = tf
if "matplotlib" in sys.modules and not hasattr(sys.modules["matplotlib"], "cm"):
    sys.modules["matplotlib"].cm = types.ModuleType("matplotlib.cm")

spec = importlib.util.spec_from_file_location("ref_util", "/root/reference/util.py")
R = importlib.util.module_from_spec(spec)
spec.loader.exec_module(R)

rng = np.random.RandomState(2024)
out = {}
stack = rng.randn(7, 40, 3).astype(np.float32)               # [frames, particles, 3]: the g_tmp stack
out["denoise_in"] = stack
for i, sigma in enumerate(((1.0, 0, 0), (0.5, 0, 0), (2.5, 0, 0))):
    out["denoise_sigma_%d" % i] = np.array(sigma, np.float64)
    out["denoise_out_%d" % i] = R.denoise(stack, sigma=sigma)
imgs = [rng.rand(37, 53, 3).astype(np.float32), rng.rand(64, 48, 4).astype(np.float32), rng.rand(20, 20, 3).astype(np.float32)]
ratios = [1.0, 0.5, 2.0, 200.0 / 200.0, 128.0 / 96.0, 53.0 / 37.0]
out["crop_ratios"] = np.array(ratios)
for i, img in enumerate(imgs):
    out["crop_in_%d" % i] = img
    for j, ra in enumerate(ratios):
        out["crop_out_%d_%d" % (i, j)] = R.crop_ratio(img, ra)
# Laplacian-pyramid smoothing kernels (util.py:27-46), as the module builds them at import time
out["k5x5"] = np.asarray(R.k5x5[1][:, :, 0, 0], np.float64)
out["k5x5x5"] = np.asarray(R.k5x5x5[1][:, :, :, 0, 0], np.float64)
out["cosine_decay"] = np.array([R.cosine_decay(s_, 20, 0.1, 2.0) for s_ in (0, 5, 10, 20, 30)], np.float64)
# small host helpers of the drivers' ``from util import *`` surface (plain NumPy in the reference): colour conversions,
# the image tiling of save_image, the boolean flag parser
c = rng.rand(3, 11).astype(np.float64)
out["rgb_in"] = c
out["rgb2yuv"] = np.stack(R.rgb2yuv(c[0], c[1], c[2]))
out["yuv2rgb"] = np.stack(R.yuv2rgb(c[0] - 0.2, c[1] - 0.5, c[2] - 0.5))
tiles = (rng.rand(5, 6, 7) * 255).astype(np.uint8)
out["grid_in"] = tiles
out["grid_gray_n2_p2"] = R.make_grid(tiles, nrow=2, padding=2)
out["grid_gray_n8_p0"] = R.make_grid(tiles, nrow=8, padding=0)
tiles3 = (rng.rand(3, 4, 5, 3) * 255).astype(np.uint8)
out["grid3_in"] = tiles3
out["grid_rgb_n2_p3"] = R.make_grid(tiles3, nrow=2, padding=3, gray=False)
words = ["true", "True", "1", "0", "false", "yes", "TRUE", ""]
out["str2bool"] = np.array([R.str2bool(w_) for w_ in words])
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "util_reference.npz")
np.savez_compressed(path, **out)
print(path, len(out), "arrays")
