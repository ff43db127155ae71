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
