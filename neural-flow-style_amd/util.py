"""Host glue of the optimisation path (reference util.py:169-207, 404-425): temporal
smoothing of the per-frame updates, style-image cropping / resizing, log-dir helpers.
Everything here is tiny NumPy/SciPy work on the host; nothing is on the GPU hot path."""
from __future__ import annotations

import json
import os

import numpy as np
from scipy.ndimage import gaussian_filter, map_coordinates


def denoise(img, sigma):
    """Gaussian filter over the frame axis of the stacked per-frame updates (util.py:169-170;
    SciPy defaults: reflect boundary, truncate at 4 sigma)."""
    return gaussian_filter(np.asarray(img), sigma=sigma)


def crop_ratio(img, ratio):
    """centre crop to width/height = ratio (util.py:176-185)"""
    h, w = img.shape[:2]
    if w / float(h) > ratio:
        hw = [h, int(h * ratio)]
    else:
        hw = [int(w / ratio), w]
    assert hw[0] <= h and hw[1] <= w
    oy, ox = int((h - hw[0]) * 0.5), int((w - hw[1]) * 0.5)
    return img[oy:oy + hw[0], ox:ox + hw[1]]


def _resize_plane(a, size, order):
    """bicubic (order 3) resize of one 2-D plane with anti-aliasing, following the published
    algorithm of skimage.transform.resize (0.14.x): Gaussian pre-filter with
    sigma = max(0, (scale-1)/2) per axis (mode 'constant'), then sampling at
    src = (dst + 0.5) * scale - 0.5 with zero padding, result clipped to the input range.
    scikit-image is not installed here; SciPy's spline interpolation stands in for its
    cubic-convolution warp (host-side style-image preparation only -- parity unpinned)."""
    a = np.asarray(a, np.float64)
    scale = [a.shape[k] / float(size[k]) for k in range(2)]
    sig = [max(0.0, (s - 1.0) / 2.0) for s in scale]
    if any(s > 0 for s in sig):
        a = gaussian_filter(a, sig, mode="constant", cval=0.0)
    yy = (np.arange(size[0]) + 0.5) * scale[0] - 0.5
    xx = (np.arange(size[1]) + 0.5) * scale[1] - 0.5
    g = np.meshgrid(yy, xx, indexing="ij")
    out = map_coordinates(a, g, order=order, mode="constant", cval=0.0, prefilter=order > 1)
    return np.clip(out, a.min(), a.max())


def resize(img, size=None, f=None, order=1):
    """util.resize (util.py:187-207): normalise to [0,1] when the range exceeds [-1,1], resize each
    channel, de-normalise."""
    img = np.asarray(img, np.float32)
    vmin, vmax = float(img.min()), float(img.max())
    norm = vmin < -1 or vmax > 1
    if norm:
        img = (img - vmin) / (vmax - vmin)
    if size is None:
        size = [int(round(img.shape[0] * f)), int(round(img.shape[1] * f))]
    size = [int(size[0]), int(size[1])]
    if img.ndim == 2:
        out = _resize_plane(img, size, order)
    else:
        out = np.stack([_resize_plane(img[..., c], size, order) for c in range(img.shape[-1])], -1)
    out = out.astype(np.float32)
    return out * (vmax - vmin) + vmin if norm else out


def prepare_dirs_and_logger(config):
    """log dir + params.json dump (util.py:404-425)"""
    os.makedirs(config.log_dir, exist_ok=True)
    tag = getattr(config, "tag", "test")
    config.log_dir = os.path.join(config.log_dir, config.dataset, tag)
    os.makedirs(config.log_dir, exist_ok=True)
    params = {k: v for k, v in vars(config).items()
              if isinstance(v, (int, float, str, bool, list, tuple, type(None)))}
    with open(os.path.join(config.log_dir, "params.json"), "w") as fp:
        json.dump(params, fp, indent=4, sort_keys=True)
    return config.log_dir


def temporal_weights(n, sigma, truncate=4.0):
    """The frame-axis Gaussian of ``denoise`` (util.py:169-170 -> scipy.ndimage.gaussian_filter, defaults: mode
    'reflect', truncate 4 sigma) written out as the n x n matrix it is: ``denoise(x, (sigma,0,..))[t] = sum_s
    W[t,s] x[s]``.  Kernel: w_k = exp(-k^2 / 2 sigma^2) / sum, |k| <= int(truncate*sigma + 0.5); 'reflect' maps an
    index outside [0,n) as (d c b a | a b c d | d c b a).  The grid-sequence stylizer needs the matrix form because
    it transports x[s] to frame t before adding it (an Eulerian field cannot be filtered in place the way
    per-particle attributes can)."""
    n = int(n)
    W = np.zeros((n, n), np.float64)
    if sigma <= 0 or n <= 1:
        return np.eye(n)
    radius = int(truncate * float(sigma) + 0.5)
    k = np.arange(-radius, radius + 1)
    w = np.exp(-0.5 / (float(sigma) ** 2) * k ** 2)
    w /= w.sum()
    for t in range(n):
        for kk, wk in zip(k, w):
            s = (t + int(kk)) % (2 * n)
            if s >= n:
                s = 2 * n - 1 - s
            W[t, s] += wk
    return W


# ---- Laplacian-pyramid gradient normalisation (util.py:27-110) -------------------------------------------------------

def lap_kernel(is_3d):
    """k5x5[1] / k5x5x5[1] of the reference (util.py:27-46), built the way its module does: outer([1,4,6,4,1]) / sum in
    2-D; floor(outer(k2, k2) * k2_i) / sum with k2 = [1, 16^(1/3), 36^(1/3), 16^(1/3), 1] in 3-D.  Pinned to the
    reference's own arrays by tests/golden/util_reference.npz."""
    if not is_3d:
        k = np.float32([1, 4, 6, 4, 1])
        k = np.outer(k, k)
        return (k / k.sum()).astype(np.float32)
    k2_ = [1, 16 ** (1 / 3), 36 ** (1 / 3), 16 ** (1 / 3), 1]
    k2 = np.float32(k2_)
    k2 = np.outer(k2, k2)
    k_ = np.floor(np.array([k2 * i for i in k2_]))
    return (k_ / k_.sum()).astype(np.float32)


def cosine_decay(global_step, decay_steps, learning_rate, factor):
    """util.py:49-53"""
    global_step = min(global_step, decay_steps)
    cos_decay = np.cos(np.pi * global_step / decay_steps)
    cos_decay = (cos_decay + 1) * 0.5 * (factor - 1) + 1
    return learning_rate * cos_decay


_LAP_K = {}


def lap_normalize(img, scale_n=3, is_3d=False, c=1):
    """Laplacian-pyramid normalisation of a gradient field on the device (util.lap_normalize, util.py:95-110): img
    [D,H,W,c] (3-D) or [H,W,c] (2-D) CUDA tensor.  scale_n = 0: img / max(mean|img|, 1e-7).  Otherwise split into scale_n
    high-pass levels + the low-pass rest (lap_split_n), divide every level by its RMS (normalize_std), merge."""
    from . import ops
    import torch
    img = img.contiguous()
    if scale_n == 0:
        return ops.normalize_mean(img, use_abs=True, eps=1e-7)
    key = (bool(is_3d), img.device)
    if key not in _LAP_K:
        _LAP_K[key] = torch.as_tensor(lap_kernel(is_3d)).to(img.device).contiguous()
    k = _LAP_K[key]
    s = 5.0 if is_3d else 4.0
    levels = []
    cur = img
    for _ in range(scale_n):                                   # lap_split_n (util.py:68-75)
        lo = ops.lap_down(cur, k)
        levels.append(ops.lap_up(lo, k, cur.shape, -s, addend=cur))   # hi = img - conv_transpose(lo, k * s)
        cur = lo
    levels.append(cur)
    levels = [ops.normalize_mean(l_, use_abs=False, eps=1e-10) for l_ in levels[::-1]]
    out = levels[0]
    for hi in levels[1:]:                                       # lap_merge (util.py:77-84)
        out = ops.lap_up(out, k, hi.shape, s, addend=hi)
    return out
