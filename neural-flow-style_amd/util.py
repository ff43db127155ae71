"""Host glue of the optimisation path (reference util.py:169-207, 404-425): temporal
smoothing of the per-frame updates, style-image cropping / resizing, log-dir helpers.
Everything here is tiny NumPy/SciPy work on the host; nothing is on the GPU hot path."""
from __future__ import annotations

import json
import os

import numpy as np
from scipy.ndimage import gaussian_filter


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


def _interp_matrix(n_in, n_out, order):
    """[n_out, n_in] weights of skimage.transform.resize's 2-D path along one axis: output pixel i samples the input at
    src = (i + 0.5) * n_in / n_out - 0.5 (resize's AffineTransform through the pixel CENTRES); order 0 nearest
    (round), 1 linear, 3 Catmull-Rom cubic convolution -- the ``bicubic_interpolation`` of skimage's ``_warp_fast``
    ("Interpolation using Catmull-Rom splines, based on the bicubic convolution algorithm" of Keys), taps
    floor(src) - 1 ... floor(src) + 2; pixels outside the image read 0 (mode='constant', cval 0)"""
    src = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / float(n_out)) - 0.5
    M = np.zeros((n_out, n_in), np.float64)
    rows = np.arange(n_out)

    def put(idx, w):
        ok = (idx >= 0) & (idx < n_in)
        np.add.at(M, (rows[ok], idx[ok]), w[ok])
    if order == 0:
        put(np.round(src).astype(np.int64), np.ones(n_out))
    elif order == 1:
        i0 = np.floor(src).astype(np.int64)
        x = src - i0
        put(i0, 1.0 - x); put(i0 + 1, x)
    else:
        i0 = np.floor(src).astype(np.int64)
        x = src - i0
        x2, x3 = x * x, x * x * x
        put(i0 - 1, -0.5 * x3 + x2 - 0.5 * x)
        put(i0, 1.5 * x3 - 2.5 * x2 + 1.0)
        put(i0 + 1, -1.5 * x3 + 2.0 * x2 + 0.5 * x)
        put(i0 + 2, 0.5 * x3 - 0.5 * x2)
    return M


def _resize_plane(a, size, order):
    """One 2-D plane through the published algorithm of skimage.transform.resize (0.14.x) as the reference calls it
    (util.py:196-203: ``mode='constant', anti_aliasing=True``): Gaussian pre-filter with sigma = max(0, (scale - 1) / 2)
    per axis (``ndi.gaussian_filter(..., mode='constant', cval=0)``), then the warp through the scale transform -- for a
    2-D image that is skimage's ``_warp_fast``: separable cubic CONVOLUTION (Catmull-Rom) for order 3, not the B-spline
    of ``ndi.map_coordinates`` (which skimage only uses for n-D volumes) --, zero outside the image, result clipped to
    the input's range (``clip=True``).  scikit-image is not installed in this image, so this restatement is pinned by
    its properties only (tests/test_util_resize_cpu.py): exact copy at equal size, partition of unity and linear
    precision in the interior; "parity unpinned" against skimage 0.14.2 itself (host-side style-image preparation)."""
    a = np.asarray(a, np.float64)
    if int(order) not in (0, 1, 3):
        raise ValueError("resize order %r: 0 (nearest), 1 (bilinear) or 3 (bicubic, what the reference passes, "
                         "styler_base.py:257-260) -- skimage's order 2 / 4 / 5 warps are not restated" % (order,))
    order = int(order)
    scale = [a.shape[k] / float(size[k]) for k in range(2)]
    sig = [max(0.0, (s - 1.0) / 2.0) for s in scale]
    if any(s > 0 for s in sig):
        a = gaussian_filter(a, sig, mode="constant", cval=0.0)
    lo, hi = a.min(), a.max()             # (skimage warps and clips the FILTERED image: the range after the pre-filter)
    out = _interp_matrix(a.shape[0], size[0], order) @ a @ _interp_matrix(a.shape[1], size[1], order).T
    return np.clip(out, lo, hi)


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
    # A level of the 200^3 x 3 pyramid is 96 MB: where the cell kernel applies, the sum of squares of a high-pass level is
    # formed by the kernel that writes it and its 1 / RMS is applied by the merge kernel that reads it (no normalisation
    # pass of its own: two reads and a write of the level less)
    fuse = os.environ.get("NFS_LAP_FUSE", "1") != "0"
    his = []
    cur = img
    for _ in range(scale_n):                                   # lap_split_n (util.py:68-75)
        lo = ops.lap_down(cur, k)
        if fuse and ops.lap_up_rms_parts(cur.shape) > 0:       # hi = img - conv_transpose(lo, k * s), + its sum of squares
            his.append(ops.lap_up_rms(lo, k, cur.shape, -s, addend=cur, want_part=True))
        else:
            his.append((ops.lap_up(lo, k, cur.shape, -s, addend=cur), None))
        cur = lo
    out = ops.normalize_mean(cur, use_abs=False, eps=1e-10)     # the low-pass rest (normalize_std, util.py:86-90)
    for hi, part in his[::-1]:                                  # lap_merge (util.py:77-84) of the normalised levels
        if part is not None:
            out = ops.lap_up_rms(out, k, hi.shape, s, addend=hi, addend_part=part, eps=1e-10)
        else:
            out = ops.lap_up(out, k, hi.shape, s, addend=ops.normalize_mean(hi, use_abs=False, eps=1e-10))
    return out


# ---- small host helpers of the reference's ``from util import *`` surface (util.py:209-235, 401-431, 484-531) --------
# NumPy / PIL only; none of them is on the hot path.  Viewers and external tools (draw_pt, draw_voxel, npz2vdb,
# save_video, v2rgb) are out of scope.

def str2bool(v):
    """argparse type of the boolean flags (util.py:401-402): 'true' / '1' in any case are True, everything else False"""
    return str(v).lower() in ("true", "1")


def get_time():
    """time stamp of the log directories (util.py:427-428)"""
    from datetime import datetime
    return datetime.now().strftime("%m%d_%H%M%S")


def save_config(config):
    """``params.json`` in ``config.log_dir`` (util.py:418-425); entries json cannot hold (the rng) are skipped"""
    path = os.path.join(config.log_dir, "params.json")
    print("[*] MODEL dir: %s" % config.log_dir)
    print("[*] PARAM path: %s" % path)
    keep = {k: v for k, v in vars(config).items() if isinstance(v, (int, float, str, bool, list, tuple, type(None)))}
    with open(path, "w") as fp:
        json.dump(keep, fp, indent=4, sort_keys=True)


def save_density(d, d_path):
    """a [H,W] image in [0,1] as a grey 8-bit picture (util.py:209-213)"""
    from PIL import Image
    g = (np.asarray(d) * 255).astype(np.uint8)
    Image.fromarray(np.repeat(g[..., None], 3, axis=-1)).save(d_path)


_YUV_FROM_RGB = np.array([[0.299, 0.587, 0.114],
                          [-0.14714119, -0.28886916, 0.43601035],
                          [0.61497538, -0.51496512, -0.10001026]])
_RGB_FROM_YUV = np.array([[1.0, 0.0, 1.13988303],
                          [1.0, -0.394642334, -0.58062185],
                          [1.0, 2.03206185, 0.0]])


def rgb2yuv(r, g, b):
    """TF's rgb_to_yuv matrix, channel by channel (util.py:229-233)"""
    m = _YUV_FROM_RGB
    return tuple(m[i, 0] * r + m[i, 1] * g + m[i, 2] * b for i in range(3))


def yuv2rgb(y, u, v):
    """its inverse as TF writes it (util.py:215-227)"""
    m = _RGB_FROM_YUV
    return tuple(m[i, 0] * y + m[i, 1] * u + m[i, 2] * v for i in range(3))


def hsv2rgb(h, s, v):
    """h, s, v in [0,1] (arrays of one shape) -> r, g, b (util.py:235-269: the six-sector construction)"""
    h, s, v = (np.asarray(a, np.float32) for a in (h, s, v))
    c = s * v
    h6 = h * 6
    x = c * (1 - np.abs(np.mod(h6, 2) - 1))
    sector = h6.astype(np.int32)
    zero = np.zeros_like(c)
    # (r, g, b) before the offset, per sector 0..5; a sector outside 0..5 (h = 1 exactly) leaves zeros, as there
    table = ((c, x, zero), (x, c, zero), (zero, c, x), (zero, x, c), (x, zero, c), (c, zero, x))
    out = [np.zeros_like(c) for _ in range(3)]
    for k, comps in enumerate(table):
        hit = sector == k
        for ch in range(3):
            out[ch] = np.where(hit, comps[ch], out[ch])
    m = v - c
    return out[0] + m, out[1] + m, out[2] + m


def make_grid(tensor, nrow=8, padding=2, normalize=False, scale_each=False, gray=True):
    """[B,H,W(,3)] uint8 images tiled ``nrow`` per row with ``padding`` black pixels between them (util.py:484-515,
    after torchvision's make_grid; ``normalize`` / ``scale_each`` are accepted and ignored, as there)"""
    t = np.asarray(tensor)
    n = t.shape[0]
    cols = min(int(nrow), n)
    rows = -(-n // cols)
    ch, cw = int(t.shape[1] + padding), int(t.shape[2] + padding)
    edge = (1 + padding // 2) if padding else 0
    shape = [ch * rows + edge, cw * cols + edge] + ([] if gray else [3])
    grid = np.zeros(shape, np.uint8)
    for k in range(n):
        y, x = divmod(k, cols)
        grid[y * ch + edge:y * ch + edge + t.shape[1], x * cw + edge:x * cw + edge + t.shape[2]] = t[k]
    return grid


def save_image(tensor, filename, nrow=8, padding=2, normalize=False, scale_each=False, single=False, gray=True):
    """a batch of uint8 images as one tiled picture, or (``single``) one image as it is (util.py:517-531)"""
    from PIL import Image
    arr = np.asarray(tensor) if single else make_grid(tensor, nrow=nrow, padding=padding, normalize=normalize,
                                                      scale_each=scale_each, gray=gray)
    Image.fromarray(arr).save(filename)
