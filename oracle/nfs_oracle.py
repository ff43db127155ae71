"""CPU oracle for the stylisation hot path of byungsook/neural-flow-style.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this
module: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and there only as the checker / timed baseline.

What it is: a function-by-function restatement, in PyTorch-CPU tensor ops with
autograd (float32 or float64), of the TensorFlow-1.15 graph the reference
builds for this path.  Every function cites the reference ``file:line`` it
follows (paths relative to the reference tree).

Pinning status -- "parity unpinned" except for the warp kernel:
  * The reference cannot be imported here (tensorflow==1.15 / tf.contrib.slim
    are absent, there is no native code to compile), and its tree holds a
    single known-answer vector: the 5x5 bilinear-warp docstring in
    transform.py:1859-1885.  ``tests/test_oracle_kat.py`` pins
    ``interpolate2d`` to it (and the 3-D twin to it by embedding).
  * Everything else is pinned only to (a) the cited lines, (b) float64
    ``torch.autograd.gradcheck`` of each operator, (c) closed-form adjoints
    (render), (d) an independent NumPy loop restatement for the splat.
  * The Inception-v1 loss network (section 8(f)-3 below): PARITY UNPINNED -- the
    GraphDef is not in the reference tree and cannot be parsed here; the
    restatement follows the published topology of that file.
  The arithmetic of the reference lives in TensorFlow 1.15 (un-vendored, pinned
  by setup.bat:8 ``pip install tensorflow==1.15``); the TF1 op semantics that
  differ from PyTorch defaults are restated explicitly below (ApplyAdam epsilon
  placement, legacy image.resize coordinates, Maximum/ReduceMax gradient ties,
  ScatterNd dropping out-of-range indices on GPU, VALID 2x2 avg-pool).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# A1  mgrid / batch_mgrid                                   transform.py:152-204
# --------------------------------------------------------------------------

def mgrid(*n, dtype=torch.float32, low=-1.0, high=1.0):
    """linspace(low, high, n_k) per axis, meshgrid(indexing='ij'), stacked on a
    leading axis -> [len(n), n0, n1, ...]   (transform.py:171-177)."""
    axes = [torch.linspace(low, high, int(k), dtype=dtype) for k in n]
    return torch.stack(torch.meshgrid(*axes, indexing="ij"))


# --------------------------------------------------------------------------
# A2  _interpolate2d / _interpolate3d                       transform.py:280-433
# --------------------------------------------------------------------------

def interpolate2d(imgs, x, y, out_shape):
    """imgs [B,X,Y,C]; x,y flat normalised coords in [-1,1] (len B*X'*Y').

    Follows transform.py:280-341: scale to index space (299-300), floor,
    x1=x0+1, clip BOTH to [0,n-1] (308-311), weights from ``x - float(clipped
    x0)`` (330-331) => exact border replication, 4 gathers + add_n (322-336).
    """
    B, X, Y, C = imgs.shape
    nb, Xo, Yo = out_shape
    x = (x + 1.0) * (X - 1.0) * 0.5
    y = (y + 1.0) * (Y - 1.0) * 0.5
    x0 = torch.floor(x).long(); x1 = x0 + 1
    y0 = torch.floor(y).long(); y1 = y0 + 1
    x0 = x0.clamp(0, X - 1); x1 = x1.clamp(0, X - 1)
    y0 = y0.clamp(0, Y - 1); y1 = y1.clamp(0, Y - 1)
    # transform.py:312 uses the OUTPUT dims for the batch offset
    base = torch.arange(nb).repeat_interleave(Xo * Yo) * (Xo * Yo)
    flat = imgs.reshape(-1, C)
    i00 = flat[base + x0 * Y + y0]; i01 = flat[base + x0 * Y + y1]
    i10 = flat[base + x1 * Y + y0]; i11 = flat[base + x1 * Y + y1]
    dx = (x - x0.to(x.dtype)).unsqueeze(1)
    dy = (y - y0.to(y.dtype)).unsqueeze(1)
    out = ((1 - dx) * (1 - dy)) * i00 + ((1 - dx) * dy) * i01 \
        + (dx * (1 - dy)) * i10 + (dx * dy) * i11
    return out.reshape(nb, Xo, Yo, C)


def interpolate3d(imgs, x, y, z, out_shape):
    """imgs [B,X,Y,Z,C]; 3-D twin, transform.py:343-433 (8 gathers)."""
    B, X, Y, Z, C = imgs.shape
    nb, Xo, Yo, Zo = out_shape
    x = (x + 1.0) * (X - 1.0) * 0.5
    y = (y + 1.0) * (Y - 1.0) * 0.5
    z = (z + 1.0) * (Z - 1.0) * 0.5
    x0 = torch.floor(x).long(); x1 = x0 + 1
    y0 = torch.floor(y).long(); y1 = y0 + 1
    z0 = torch.floor(z).long(); z1 = z0 + 1
    x0 = x0.clamp(0, X - 1); x1 = x1.clamp(0, X - 1)
    y0 = y0.clamp(0, Y - 1); y1 = y1.clamp(0, Y - 1)
    z0 = z0.clamp(0, Z - 1); z1 = z1.clamp(0, Z - 1)
    base = torch.arange(nb).repeat_interleave(Xo * Yo * Zo) * (Xo * Yo * Zo)
    flat = imgs.reshape(-1, C)
    dx = (x - x0.to(x.dtype)).unsqueeze(1)
    dy = (y - y0.to(y.dtype)).unsqueeze(1)
    dz = (z - z0.to(z.dtype)).unsqueeze(1)
    out = 0
    for xi, wx in ((x0, 1 - dx), (x1, dx)):
        for yi, wy in ((y0, 1 - dy), (y1, dy)):
            for zi, wz in ((z0, 1 - dz), (z1, dz)):
                out = out + (wx * wy * wz) * flat[base + xi * (Y * Z) + yi * Z + zi]
    return out.reshape(nb, Xo, Yo, Zo, C)


def batch_warp2d(imgs, mappings, out_shape):
    """mappings [B,2,X,Y] (transform.py:206-236)."""
    nb = out_shape[0]
    c = mappings.reshape(nb, 2, -1)
    return interpolate2d(imgs, c[:, 0].reshape(-1), c[:, 1].reshape(-1), out_shape)


FAST_WARP = False  # set by bench.py's cpu_baseline only: multi-threaded F.grid_sample, proven identical to
                   # the 8-gather restatement in tests/test_oracle_kat.py (fwd 1e-12, both gradients)


def batch_warp3d(imgs, mappings, out_shape):
    """mappings [B,3,X,Y,Z] (transform.py:238-269)."""
    if FAST_WARP and imgs.shape[0] == mappings.shape[0]:
        grid = torch.stack([mappings[:, 2], mappings[:, 1], mappings[:, 0]], -1)
        out = F.grid_sample(imgs.permute(0, 4, 1, 2, 3), grid, mode="bilinear", padding_mode="border",
                            align_corners=True)
        return out.permute(0, 2, 3, 4, 1)
    nb = out_shape[0]
    c = mappings.reshape(nb, 3, -1)
    return interpolate3d(imgs, c[:, 0].reshape(-1), c[:, 1].reshape(-1),
                         c[:, 2].reshape(-1), out_shape)


def affine_warp2d(imgs, theta):
    """batch_affine_warp2d (transform.py:435-470) -- only used by the KAT."""
    B, X, Y, _ = imgs.shape
    th = theta.reshape(-1, 2, 3)
    g = mgrid(X, Y, dtype=imgs.dtype).reshape(1, 2, -1).expand(B, -1, -1)
    tg = th[:, :, :2] @ g + th[:, :, 2:]
    return batch_warp2d(imgs, tg.reshape(B, 2, X, Y), [B, X, Y])


# --------------------------------------------------------------------------
# A3  rotate                                                transform.py:611-628
# --------------------------------------------------------------------------

def rotate(d, rot_mat):
    """d [B,D,H,W,C], rot_mat [V,3,3] -> [B*V,D,H,W,C].

    tile d V times (620), coords = R @ mgrid (621-625), trilinear warp (627).
    """
    B, D, H, W, C = d.shape
    V = rot_mat.shape[0]
    nb = B * V
    dd = d.repeat(V, 1, 1, 1, 1)
    r = rot_mat.to(d.dtype).repeat(B, 1, 1)
    g = mgrid(D, H, W, dtype=d.dtype).reshape(1, 3, -1).expand(nb, -1, -1)
    g = (r @ g).reshape(nb, 3, D, H, W)
    return batch_warp3d(dd, g, [nb, D, H, W])


# --------------------------------------------------------------------------
# A11  advect (order 1)                                     transform.py:557-569
# --------------------------------------------------------------------------

def advect(d, vel):
    """d [1,D,H,W,C], vel [1,D,H,W,3] in normalised units; x' = x - v (566)."""
    assert d.shape[0] == 1  # n_batch hard-wired to 1 (558)
    _, D, H, W, _ = d.shape
    g = mgrid(D, H, W, dtype=d.dtype).unsqueeze(0) - vel.permute(0, 4, 1, 2, 3)
    return batch_warp3d(d, g, [1, D, H, W])


def advect2d(d, vel):
    """2-D branch, transform.py:583-588."""
    assert d.shape[0] == 1
    _, H, W, _ = d.shape
    g = mgrid(H, W, dtype=d.dtype).unsqueeze(0) - vel.permute(0, 3, 1, 2)
    return batch_warp2d(d, g, [1, H, W])


def _stencil_extrema(d, coords):
    """min / max of d over the (clipped) corners of the interpolation cell of each coordinate -- what the reference's
    2x2(x2) max-pool of d sampled at the back-traced cell stands for (transform.py:574-579, 594-601).
    d [1,*dims,C], coords [1,nd,*dims] normalised."""
    dims = d.shape[1:-1]
    nd = len(dims)
    idx = []
    for k in range(nd):
        x = (coords[0, k] + 1.0) * (dims[k] - 1.0) * 0.5
        x0 = torch.floor(x).long()
        idx.append((x0.clamp(0, dims[k] - 1), (x0 + 1).clamp(0, dims[k] - 1)))
    lo = hi = None
    import itertools
    for corner in itertools.product((0, 1), repeat=nd):
        v = d[0][tuple(idx[k][corner[k]] for k in range(nd))]
        lo = v if lo is None else torch.minimum(lo, v)
        hi = v if hi is None else torch.maximum(hi, v)
    return lo[None], hi[None]


def advect_maccormack(d, vel):
    """advect order 2 (transform.py:570-582 3-D, 590-607 2-D), the scheme those lines transcribe with the limiter done
    as intended (the reference's own limiter cannot run: tf.to_int32 of [-1,1] coordinates, ``d_max[grids]``):
    d_fwd = SL(d, v); d_bwd = SL(d_fwd, -v) (grids_ = mgrid + vel); d_adv = d_fwd + (d - d_bwd)/2; soft clamp: where
    d_adv > max or < min of d over the back-traced cell's corners, keep d_fwd."""
    is_3d = d.dim() == 5
    if is_3d:
        _, D, H, W, _ = d.shape
        g = mgrid(D, H, W, dtype=d.dtype).unsqueeze(0)
        vp = vel.permute(0, 4, 1, 2, 3)
        d_fwd = batch_warp3d(d, g - vp, [1, D, H, W])
        d_bwd = batch_warp3d(d_fwd, g + vp, [1, D, H, W])
    else:
        _, H, W, _ = d.shape
        g = mgrid(H, W, dtype=d.dtype).unsqueeze(0)
        vp = vel.permute(0, 3, 1, 2)
        d_fwd = batch_warp2d(d, g - vp, [1, H, W])
        d_bwd = batch_warp2d(d_fwd, g + vp, [1, H, W])
    d_adv = d_fwd + (d - d_bwd) * 0.5
    lo, hi = _stencil_extrema(d, g - vp)
    return torch.where((d_adv > hi) | (lo > d_adv), d_fwd, d_adv)


def curl(s, is_2d=True):
    """transform.py:517-555, line by line (forward differences, last slice repeated)."""
    if is_2d:
        u = s[:, 1:, :, 0] - s[:, :-1, :, 0]
        v = s[:, :, :-1, 0] - s[:, :, 1:, 0]
        u = torch.cat([u, u[:, -1:, :]], dim=1)
        v = torch.cat([v, v[:, :, -1:]], dim=2)
        return torch.stack([u, v], dim=-1)
    dvdx = s[:, :, :, 1:, 1] - s[:, :, :, :-1, 1]
    dwdx = s[:, :, :, 1:, 2] - s[:, :, :, :-1, 2]
    dudy = s[:, :, 1:, :, 0] - s[:, :, :-1, :, 0]
    dwdy = s[:, :, 1:, :, 2] - s[:, :, :-1, :, 2]
    dudz = s[:, 1:, :, :, 0] - s[:, :-1, :, :, 0]
    dvdz = s[:, 1:, :, :, 1] - s[:, :-1, :, :, 1]
    dvdx = torch.cat((dvdx, dvdx[:, :, :, -1:]), dim=3)
    dwdx = torch.cat((dwdx, dwdx[:, :, :, -1:]), dim=3)
    dudy = torch.cat((dudy, dudy[:, :, -1:, :]), dim=2)
    dwdy = torch.cat((dwdy, dwdy[:, :, -1:, :]), dim=2)
    dudz = torch.cat((dudz, dudz[:, -1:, :, :]), dim=1)
    dvdz = torch.cat((dvdz, dvdz[:, -1:, :, :]), dim=1)
    return torch.stack([dwdy - dvdz, dudz - dwdx, dvdx - dudy], dim=-1)


def _tf_same_pads(n, window=5, stride=2):
    out = -(-n // stride)
    total = max((out - 1) * stride + window - n, 0)
    return total // 2, total - total // 2


def lap_normalize(img, k, scale_n=3):
    """util.lap_normalize (util.py:95-110) restated with torch's convolutions: img [D,H,W,C] or [H,W,C], k the
    reference's k5x5x5[1] / k5x5[1] kernel ([5,5,5] / [5,5]).  TF 'SAME' geometry is written out as explicit
    asymmetric zero padding; conv_transpose = the exact adjoint of that padded strided correlation."""
    nd = img.dim() - 1
    if scale_n == 0:
        return img / torch.clamp(img.abs().mean(), min=1e-7)
    C = img.shape[-1]
    kk = torch.as_tensor(k, dtype=img.dtype)
    s = 5.0 if nd == 3 else 4.0
    conv = F.conv3d if nd == 3 else F.conv2d
    w = kk.reshape((1, 1) + tuple(kk.shape)).repeat(C, 1, *([1] * nd))      # depthwise: the same k per channel

    def down(x):                                                            # x [*dims, C]
        xc = x.permute(nd, *range(nd)).unsqueeze(0)
        pads = []
        for n in reversed(x.shape[:nd]):
            pads += list(_tf_same_pads(n))
        return conv(F.pad(xc, pads), w, stride=2, groups=C)[0].permute(*range(1, nd + 1), 0)

    def up(lo, shape):                                                      # adjoint of ``down`` at input shape
        probe = torch.zeros(shape, dtype=lo.dtype, requires_grad=True)
        (g,) = torch.autograd.grad(down(probe), probe, lo)
        return g

    levels, cur = [], img
    for _ in range(scale_n):
        lo = down(cur)
        levels.append(cur - s * up(lo, cur.shape))
        cur = lo
    levels.append(cur)
    levels = [l_ / torch.clamp(torch.sqrt((l_ ** 2).mean()), min=1e-10) for l_ in levels[::-1]]
    out = levels[0]
    for hi in levels[1:]:
        out = s * up(out, hi.shape) + hi
    return out


def transport(g, v, a, b, recursive=True):
    """StylerBase._transport_tf (styler_base.py:76-89): move field g from frame
    a to frame b through the per-frame velocities v[F,D,H,W,3]."""
    if a < b:
        if recursive:
            for i in range(a, b):
                g = advect(g, v[i:i + 1])
        else:
            g = advect(g, v[a:a + 1] * (b - a))
    elif a > b:
        if recursive:
            for i in reversed(range(b, a)):
                g = advect(g, -v[i:i + 1])
        else:
            g = advect(g, -v[a - 1:a] * (a - b))
    return g


# --------------------------------------------------------------------------
# A9  smoothing conv + clamp                                styler_3p.py:112-125
# --------------------------------------------------------------------------

class _TFMaximum0(torch.autograd.Function):
    """tf.maximum(d, 0): TF routes the gradient to ``d`` where d >= 0 (ties go
    to the first argument), unlike torch.relu (zero at 0)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.clamp_min(x, 0)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * (x >= 0).to(g.dtype)


def smooth_kernel3d(k, dtype=torch.float32):
    """k1=[1,k,1]; k3 = k1 (x) k1 (x) k1 / sum  (styler_3p.py:114-120)."""
    k1 = torch.tensor([1.0, float(k), 1.0], dtype=dtype)
    k3 = torch.einsum("i,j,k->ijk", k1, k1, k1)
    return k3 / k3.sum()


def smooth3d_relu(d, k):
    """d [B,D,H,W,1] -> conv3d SAME (zero pad) with the (k+2)^-3 kernel, then
    max(.,0)  (styler_3p.py:112-125).  k<=0 skips the conv."""
    if k > 0:
        w = smooth_kernel3d(k, d.dtype)[None, None]
        x = d.permute(0, 4, 1, 2, 3)
        x = F.conv3d(x, w, padding=1)
        d = x.permute(0, 2, 3, 4, 1)
    return _TFMaximum0.apply(d)


# --------------------------------------------------------------------------
# A4  render                                                styler_3p.py:147-158
# --------------------------------------------------------------------------

def render(d, transmit, liquid=False):
    """d [B,D,H,W,1] -> [B,H,W,1].

    smoke: T = exp(-tau*cumsum(d[:,::-1]))[:,::-1] (155); I = sum(d*T, axis=1)
    (156-157); I /= max over the WHOLE tensor (158, gradient flows through
    the max, ties split equally as TF's reduce_max gradient does).
    liquid: 1 - exp(-tau * sum_z d) (150-152).
    ``liquid`` = 'max' / 'mean' (north_star's ray modes; not in the mounted reference beyond the commented-out
    ``d = tf.reduce_max(d, axis=1)`` of line 149): reduce_max / reduce_mean along the ray, un-normalised like the liquid
    form; torch's amax splits its gradient equally among ties, as TF's reduce_max does.
    """
    if liquid == "max":
        return d.amax(dim=1)
    if liquid == "mean":
        return d.mean(dim=1)
    if liquid:
        tr = torch.exp(-torch.cumsum(d.flip(1), dim=1) * transmit)
        return 1 - tr[:, -1]
    tr = torch.exp(-torch.cumsum(d.flip(1), dim=1) * transmit).flip(1)
    img = (d * tr).sum(dim=1)
    return img / img.amax()


def render_unnormalised(d, transmit):
    tr = torch.exp(-torch.cumsum(d.flip(1), dim=1) * transmit).flip(1)
    return (d * tr).sum(dim=1)


def render_adjoint_closed_form(d, transmit, g_img):
    """dI/dd[k] = T[k] - tau * sum_{z<=k} d[z] T[z]; used to cross-check
    autograd of the un-normalised integral."""
    tr = torch.exp(-torch.cumsum(d.flip(1), dim=1) * transmit).flip(1)
    pre = torch.cumsum(d * tr, dim=1)
    return (tr - transmit * pre) * g_img.unsqueeze(1)


# --------------------------------------------------------------------------
# A5  _plugin_to_loss_net                     styler_base.py:33-45, vgg.py:50-53
# --------------------------------------------------------------------------

VGG_MEAN = (0.485 * 255, 0.456 * 255, 0.406 * 255)  # vgg.py:18-20


def tf1_resize_bilinear(x, oh, ow):
    """tf.compat.v1.image.resize(BILINEAR): align_corners=False, no half-pixel
    centres: src = dst * in/out; x1 = min(x0+1, in-1).   x [B,H,W,C]."""
    B, H, W, C = x.shape
    sy = torch.arange(oh, dtype=x.dtype) * (H / oh)
    sx = torch.arange(ow, dtype=x.dtype) * (W / ow)
    y0 = torch.floor(sy).long(); y1 = torch.clamp(y0 + 1, max=H - 1)
    x0 = torch.floor(sx).long(); x1 = torch.clamp(x0 + 1, max=W - 1)
    ly = (sy - y0.to(x.dtype)).view(1, oh, 1, 1)
    lx = (sx - x0.to(x.dtype)).view(1, 1, ow, 1)
    top = x[:, y0][:, :, x0] * (1 - lx) + x[:, y0][:, :, x1] * lx
    bot = x[:, y1][:, :, x0] * (1 - lx) + x[:, y1][:, :, x1] * lx
    return top * (1 - ly) + bot * ly


_TF_BICUBIC_TABLE = 1024


def _tf1_bicubic_axis(n_in, n_out, dtype):
    """Indices/weights of TF's legacy ResizeBicubic (A=-0.75, 1024-entry
    coefficient table, align_corners=False, half_pixel_centers=False)."""
    a = -0.75
    t = np.arange(_TF_BICUBIC_TABLE + 1, dtype=np.float32) / np.float32(_TF_BICUBIC_TABLE)
    c0 = ((a + 2) * t - (a + 3)) * t * t + 1
    t1 = t + 1
    c1 = ((a * t1 - 5 * a) * t1 + 8 * a) * t1 - 4 * a
    scale = np.float32(n_in) / np.float32(n_out)
    src = np.arange(n_out, dtype=np.float32) * scale
    loc = np.floor(src).astype(np.int64)
    off = np.rint((src - loc) * _TF_BICUBIC_TABLE).astype(np.int64)
    w = np.stack([c1[off], c0[off], c0[_TF_BICUBIC_TABLE - off], c1[_TF_BICUBIC_TABLE - off]], 1)
    idx = np.clip(np.stack([loc - 1, loc, loc + 1, loc + 2], 1), 0, n_in - 1)
    return torch.from_numpy(idx), torch.from_numpy(w.astype(np.float64)).to(dtype)


def tf1_resize_bicubic(x, oh, ow):
    """tf.compat.v1.image.resize(BICUBIC) on [B,H,W,C] (styler_base.py:166)."""
    B, H, W, C = x.shape
    iy, wy = _tf1_bicubic_axis(H, oh, x.dtype)
    ix, wx = _tf1_bicubic_axis(W, ow, x.dtype)
    rows = (x[:, iy] * wy.view(1, oh, 4, 1, 1)).sum(2)          # [B,oh,W,C]
    return (rows[:, :, ix] * wx.view(1, 1, ow, 4, 1)).sum(3)    # [B,oh,ow,C]


def plugin_to_loss_net(d, resize_scale=1.0, is_color=False):
    """[B,H,W,1 or 3] in [0,1] -> d_img [B,H',W',3] in 0..255
    (styler_base.py:33-45).  Mean subtraction happens inside vgg (vgg.py:51)."""
    if not np.isclose(resize_scale, 1):
        h = int(np.float32(resize_scale) * np.float32(d.shape[1]))
        w = int(np.float32(resize_scale) * np.float32(d.shape[2]))
        d = tf1_resize_bilinear(d, h, w)
    d = d * 255
    if not is_color:
        d = torch.cat([d] * 3, dim=-1)
    return d


# --------------------------------------------------------------------------
# A6  VGG-19 (slim, avg-pool)                                     vgg.py:89-120
# --------------------------------------------------------------------------

VGG19_CFG = (("conv1", 2, 64), ("conv2", 2, 128), ("conv3", 4, 256),
             ("conv4", 4, 512), ("conv5", 4, 512))


def vgg19_layer_names(upto="conv5_1"):
    names = []
    for blk, reps, _ in VGG19_CFG:
        for i in range(reps):
            names.append("%s_%d" % (blk, i + 1))
            if names[-1] == upto:
                return names
    return names


def synthetic_vgg19_weights(seed=123, upto="conv5_1", dtype=np.float32, width_div=1):
    """Seeded He-normal stand-in for vgg_19_2016_08_28 (not shipped, no
    network): w ~ N(0, 2/(9 Cin)) HWIO [3,3,Cin,Cout], b ~ 0.01 N(0,1), drawn
    layer by layer from RandomState(seed)."""
    rng = np.random.RandomState(seed)
    out = OrderedDict()
    cin = 3
    for blk, reps, cout in VGG19_CFG:
        cout = max(cout // width_div, 4)
        for i in range(reps):
            name = "%s_%d" % (blk, i + 1)
            w = rng.randn(3, 3, cin, cout) * math.sqrt(2.0 / (9 * cin))
            b = rng.randn(cout) * 0.01
            out[name] = (w.astype(dtype), b.astype(dtype))
            cin = cout
            if name == upto:
                return out
    return out


def vgg19_features(d_img, weights, upto="conv5_1"):
    """d_img [B,H,W,3] 0..255 -> OrderedDict name -> [B,h,w,C] post-ReLU.

    preprocess: subtract RGB mean only (vgg.py:50-53); conv 3x3 SAME stride 1 +
    bias + ReLU (vgg.py:44-48, slim.conv2d defaults); slim.avg_pool2d [2,2]
    => stride 2, VALID (odd sizes floor) (vgg.py:93-104); end points are the
    post-ReLU conv outputs (vgg.py:55-66).
    """
    x = d_img - torch.tensor(VGG_MEAN, dtype=d_img.dtype)
    x = x.permute(0, 3, 1, 2)
    feats = OrderedDict()
    for blk, reps, _ in VGG19_CFG:
        for i in range(reps):
            name = "%s_%d" % (blk, i + 1)
            w, b = weights[name]
            w = torch.as_tensor(w, dtype=x.dtype).permute(3, 2, 0, 1)  # HWIO -> OIHW
            x = F.relu(F.conv2d(x, w, torch.as_tensor(b, dtype=x.dtype), padding=1))
            feats[name] = x.permute(0, 2, 3, 1)
            if name == upto:
                return feats
        x = F.avg_pool2d(x, 2, 2)  # VALID, floor
        feats["pool" + blk[-1]] = x.permute(0, 2, 3, 1)
    return feats


# --------------------------------------------------------------------------
# 8(f)-3  Inception-v1 ("inception5h", tensorflow_inception_graph.pb)
#                                          styler_base.py:17-30, 51-57, 91-94
# --------------------------------------------------------------------------
# The reference imports the GraphDef and fetches tensors by node name.  The file is not in the reference tree (it is
# downloaded by setup.bat) and TensorFlow is not in this image: PARITY UNPINNED.  What is restated is the published
# topology of that file -- node names, SAME padding, 3x3 max pools, two LRNs, four-branch modules concatenated in the
# order (1x1, 3x3, 5x5, pool_reduce) -- with the TF op semantics written out (asymmetric SAME padding, max pool over
# in-range taps only, tf.nn.lrn's depth window).  Channel widths come from the weights handed in.

INCEPTION_UNITS = ([("conv", "conv2d0"), ("maxpool", "maxpool0", 2), ("lrn", "localresponsenorm0"),
                    ("conv", "conv2d1"), ("conv", "conv2d2"), ("lrn", "localresponsenorm1"), ("maxpool", "maxpool1", 2),
                    ("mixed", "mixed3a"), ("mixed", "mixed3b"), ("maxpool", "maxpool4", 2)]
                   + [("mixed", "mixed4" + c) for c in "abcde"]
                   + [("maxpool", "maxpool10", 2), ("mixed", "mixed5a"), ("mixed", "mixed5b"),
                      ("avgpool", "avgpool0", 7), ("fc", "softmax2_pre_activation")])
INCEPTION_LRN = (5, 2.0, 1e-4, 0.5)          # depth_radius, bias, alpha, beta of both LRN nodes of the 5h graph


def _tf_same(n, k, stride):
    out = -(-n // stride)
    total = max((out - 1) * stride + k - n, 0)
    return total // 2, total - total // 2


def tf_conv2d_same(x, w_hwio, b=None, stride=1):
    """tf.nn.conv2d(padding='SAME') + BiasAdd on NCHW x with HWIO filters (padding split low = total // 2)"""
    kh, kw = w_hwio.shape[0], w_hwio.shape[1]
    pt, pb = _tf_same(x.shape[2], kh, stride)
    pl, pr = _tf_same(x.shape[3], kw, stride)
    w = torch.as_tensor(w_hwio, dtype=x.dtype).permute(3, 2, 0, 1)
    return F.conv2d(F.pad(x, (pl, pr, pt, pb)), w, None if b is None else torch.as_tensor(b, dtype=x.dtype), stride=stride)


def tf_maxpool3_same(x, stride):
    """tf.nn.max_pool(ksize 3, SAME): padding taps do not take part (= -inf padding)"""
    pt, pb = _tf_same(x.shape[2], 3, stride)
    pl, pr = _tf_same(x.shape[3], 3, stride)
    return F.max_pool2d(F.pad(x, (pl, pr, pt, pb), value=float("-inf")), 3, stride)


def tf_lrn(x, depth_radius, bias, alpha, beta):
    """tf.nn.lrn on NCHW: x / (bias + alpha * sum_{|j-c| <= r} x_j^2)^beta"""
    sq = F.pad(x * x, (0, 0, 0, 0, depth_radius, depth_radius))
    s = sum(sq[:, j:j + x.shape[1]] for j in range(2 * depth_radius + 1))
    return x / (bias + alpha * s) ** beta


def inception_v1_features(d_img, weights, upto, lrn=None, pool1=False):
    """d_img [B,H,W,3] 0..255 -> OrderedDict node name -> [B,h,w,C] for every addressable tensor down to the unit
    that produces ``upto``: ``U_pre_relu`` (BiasAdd) and ``U`` (Relu) of every convolution unit, the pools, the LRNs,
    the module outputs.  Input = vgg.preprocess(d) (styler_base.py:54: the VGG mean is what the reference subtracts
    for this network too); ``pool1``: first stride 1 (styler_base.py:28-30)."""
    lrn = dict(lrn or {})
    x = (d_img - torch.tensor(VGG_MEAN, dtype=d_img.dtype)).permute(0, 3, 1, 2)
    feats = OrderedDict()

    def put(name, t):
        feats[name] = t.permute(0, 2, 3, 1)

    def conv(name, t, stride=1):
        w, b = weights[name]
        pre = tf_conv2d_same(t, w, b, stride)
        put(name + "_pre_relu", pre)
        y = F.relu(pre)
        put(name, y)
        return y

    base = upto[:-len("_pre_relu")] if upto.endswith("_pre_relu") else upto
    for u in INCEPTION_UNITS:
        kind, name = u[0], u[1]
        if kind == "conv":
            x = conv(name, x, (1 if pool1 else 2) if name == "conv2d0" else 1)
        elif kind == "maxpool":
            x = tf_maxpool3_same(x, u[2])
            put(name, x)
        elif kind == "lrn":
            x = tf_lrn(x, *lrn.get(name, INCEPTION_LRN))
            put(name, x)
        elif kind == "avgpool":                      # AvgPool 7x7, stride 1, VALID
            x = F.avg_pool2d(x, u[2], 1)
            put(name, x)
        elif kind == "fc":                           # reshape [-1, 1024] -> MatMul + BiasAdd: a 1x1 convolution of the map
            w2, b2 = weights["softmax2"]
            x = tf_conv2d_same(x, np.asarray(w2).reshape((1, 1) + np.asarray(w2).shape[-2:]), b2)
            put(name, x)
        else:
            b1 = conv(name + "_1x1", x)
            b3 = conv(name + "_3x3", conv(name + "_3x3_bottleneck", x))
            b5 = conv(name + "_5x5", conv(name + "_5x5_bottleneck", x))
            pool = tf_maxpool3_same(x, 1)
            put(name + "_pool", pool)
            bp = conv(name + "_pool_reduce", pool)
            x = torch.cat([b1, b3, b5, bp], dim=1)
            put(name, x)
        if base == name or (kind == "mixed" and base.startswith(name + "_")):
            return feats
    raise KeyError(upto)


def content_target_feature(content_img, weights, layer, cfg=None, top_k=0):
    """_content_feature (styler_base.py:232-247): the content image fed at d_img, ``layer`` fetched; with ``top_k`` > 0
    (asserted to be the classifier's logits, line 241) every row of the fetched [rows, classes] array keeps its k
    largest |values| and the rest is zeroed (242-245)"""
    feats = loss_net_features(content_img, weights, layer, cfg)
    f = feats[layer].detach().clone()
    if top_k and top_k > 0:
        assert "softmax2_pre_activation" in layer
        rows = f.reshape(-1, f.shape[-1])
        idx = rows.abs().topk(int(top_k), dim=1).indices
        keep = torch.zeros_like(rows, dtype=torch.bool)
        keep.scatter_(1, idx, True)
        f = (rows * keep).reshape(f.shape)
    return f


def inception_last_layer(layers):
    """the tensor of ``layers`` whose unit comes last in the graph"""
    def unit_index(n):
        b = n[:-len("_pre_relu")] if n.endswith("_pre_relu") else n
        for i, u in enumerate(INCEPTION_UNITS):
            if b == u[1] or (u[0] == "mixed" and b.startswith(u[1] + "_")):
                return i
        raise KeyError(n)
    return max(layers, key=unit_index)


def loss_net_features(d_img, weights, upto, cfg=None):
    """the loss network ``cfg['network']`` selects (styler_base.py:47-57): VGG-19 unless it names the Inception graph"""
    cfg = cfg or {}
    if "inception" in str(cfg.get("network", "vgg")):
        return inception_v1_features(d_img, weights, upto, cfg.get("lrn"), cfg.get("pool1", False))
    return vgg19_features(d_img, weights, upto)


# --------------------------------------------------------------------------
# A7  Gram + style loss                           styler_base.py:96-102, 152-185
# --------------------------------------------------------------------------

def gram_matrix(x, batch_size=None):
    """x [B,h,w,C] -> [B',C,C], G = F^T F with F = reshape(x[i], (hw, C)); the
    reference loops range(batch_size) only (styler_base.py:98)."""
    nb = x.shape[0] if batch_size is None else batch_size
    g = []
    for i in range(nb):
        f = x[i].reshape(-1, x.shape[-1])
        g.append(f.t() @ f)
    return torch.stack(g)


def style_loss(feats, style_feats, layers, w_layers, w_style=1.0, batch_size=None,
               d_gray=None, style_mask_on_ref=False):
    """sum_l w_l * sum((G/denom - Gs/denom_s)^2) * w_style
    (styler_base.py:152-185).  denom = 2*h*w*C (157,162).  With ``d_gray``
    (style_mask=True): feature *= bicubic_resize(d_gray) and
    denom = 2*area*C (165-173)."""
    total = 0
    per_layer = []
    for name, wl in zip(layers, w_layers):
        f = feats[name]; s = style_feats[name]
        denom = 2.0 * f.shape[1] * f.shape[2] * f.shape[3]
        sdenom = 2.0 * s.shape[1] * s.shape[2] * s.shape[3]
        if d_gray is not None:
            m = tf1_resize_bicubic(d_gray, f.shape[1], f.shape[2])
            f = f * m
            area = m[..., 0].sum(dim=(1, 2), keepdim=True)
            denom = 2 * area * f.shape[3]
            if style_mask_on_ref:
                s = s * m
                sdenom = denom
        g = gram_matrix(f, batch_size) / denom
        gs = gram_matrix(s, batch_size) / sdenom
        ll = ((g - gs) ** 2).sum()
        per_layer.append(ll)
        total = total + wl * ll
    return total * w_style, per_layer


def content_loss(feature, content_channel=0, content_feature=None, w_content_amp=100.0):
    """content term of _loss (styler_base.py:135-150), before the w_content factor:
    with a content image  mean((feature - content_feature*amp)^2)  (139-142);
    else with content_channel c != 0  -mean(f[...,c]) + mean|f[...,:c]| + mean|f[...,c+1:]|  (144-147)
    (an empty slice is skipped here; TF's reduce_mean of it is NaN);
    else  -mean(feature)  (149).  ``feature`` [B,h,w,C] is one layer of the loss network."""
    if content_feature is not None:
        return ((feature - content_feature * w_content_amp) ** 2).mean()
    if content_channel:
        c = int(content_channel)
        loss = -feature[..., c].mean()
        if c > 0:
            loss = loss + feature[..., :c].abs().mean()
        if c + 1 < feature.shape[-1]:
            loss = loss + feature[..., c + 1:].abs().mean()
        return loss
    return -feature.mean()


# --------------------------------------------------------------------------
# A12  TV loss                                            styler_base.py:211-213
# --------------------------------------------------------------------------

def histogram_match(source, template, hist_bins=255):
    """util.histogram_match_tf (util.py:317-399) for one channel, restated in NumPy float32 (the graph dtype):
    255 fixed-width bins over [min, max] of source AND template (323-327); hist_range = bin centres
    range(min, max, delta) + delta/2 (329-334); s_hist / t_hist = tf.histogram_fixed_width (337-347:
    index = floor(nbins * (v - min) / (max - min)) clipped to nbins-1); quantiles = cumsum / total (352-356);
    nearest_indices = round(interp1d(t_quantiles, arange(nbins), fill (0, nbins-1))(s_quantiles)) (358-362);
    matched = hist_range[nearest_indices[clip(int((source - min) / delta), 0, nbins-1)]] (370-379).
    Returns the matched array (same shape as source).  No gradient flows through it (py_func / integer casts /
    tf.range): d loss / d source = 2 (source - matched).

    Nothing to match -- an empty source (every pixel masked out) or a flat channel (max == min over source and
    template) -- is outside the reference's domain: its tf.range(min, max, 0) / 0-count division have no defined result.
    The build DEFINES that case as "the channel is skipped": matched = source (loss 0, gradient 0)."""
    from scipy.interpolate import interp1d
    src = np.asarray(source, np.float32).reshape(-1)
    tpl = np.asarray(template, np.float32).reshape(-1)
    if src.size == 0:
        return src.reshape(np.shape(source))
    vmax = np.float32(max(src.max(), tpl.max()))
    vmin = np.float32(min(src.min(), tpl.min()))
    if not vmax > vmin:
        return src.reshape(np.shape(source)).copy()
    delta = np.float32((vmax - vmin) / np.float32(hist_bins))
    hist_range = (vmin + delta * np.arange(hist_bins, dtype=np.float32)).astype(np.float32) + delta / np.float32(2)

    def fixed_width(v):
        scaled = (v - vmin) / (vmax - vmin)
        idx = np.floor(np.float32(hist_bins) * scaled).astype(np.int64)
        return np.bincount(np.clip(idx, 0, hist_bins - 1), minlength=hist_bins)

    s_q = np.cumsum(fixed_width(src)).astype(np.float64); s_q /= s_q[-1]
    t_q = np.cumsum(fixed_width(tpl)).astype(np.float64); t_q /= t_q[-1]
    f = interp1d(t_q, np.arange(hist_bins), bounds_error=False, fill_value=(0, hist_bins - 1))
    with np.errstate(invalid="ignore", divide="ignore"):
        nearest = np.round(np.nan_to_num(f(s_q))).astype(np.int64)   # 0/0 at a flat start of the template CDF -> bin 0
    s_bin = np.clip(((src - vmin) / delta).astype(np.int64), 0, hist_bins - 1)
    return hist_range[nearest[s_bin]].reshape(np.shape(source))


def hist_loss(feature, hist_feature, mask=None):
    """histogram term of _loss (styler_base.py:187-209, as intended: the template is the fed ``hist_feature`` of the
    same layer -- the mounted line 203 reads the stale ``style_feature`` of the style loop):
    sum over images i < batch and channels j of sum((feature[i,...,j] - matched)^2) with
    matched = histogram_match(feature[i,...,j], hist_feature[i,...,j]) held constant.  feature [B,h,w,C] torch tensor
    (differentiable), hist_feature [Bt,ht,wt,C].
    ``mask`` [B,h,w,1] (the masked branch, _hist_match 104-125 + 196-201): ``tf.boolean_mask(s_, mask != 0)`` removes
    the masked-out pixels from the source before the match; the loss sums (matched - masked source)^2."""
    f = feature.detach().cpu().numpy()
    t = np.asarray(hist_feature.detach().cpu().numpy() if torch.is_tensor(hist_feature) else hist_feature)
    mk = None if mask is None else (np.asarray(mask.detach().cpu().numpy() if torch.is_tensor(mask) else mask)
                                    .reshape(f.shape[:-1]) != 0)
    m = f.copy()                                   # outside the mask: matched := source (no loss, no gradient)
    for i in range(f.shape[0]):
        for j in range(f.shape[-1]):
            tpl = t[min(i, t.shape[0] - 1), ..., j]
            if mk is None:
                m[i, ..., j] = histogram_match(f[i, ..., j], tpl)
            else:
                sel = mk[i]
                m[i, ..., j][sel] = histogram_match(f[i, ..., j][sel], tpl)
    return ((feature - torch.as_tensor(m, dtype=feature.dtype)) ** 2).sum()


def tv_loss(d_img):
    """reduce_mean over batch of tf.image.total_variation: sum|dh| + sum|dw|."""
    dh = (d_img[:, 1:] - d_img[:, :-1]).abs().sum(dim=(1, 2, 3))
    dw = (d_img[:, :, 1:] - d_img[:, :, :-1]).abs().sum(dim=(1, 2, 3))
    return (dh + dw).mean()


# --------------------------------------------------------------------------
# A8  SPH splat: W, p2g, p2g_wavg     transform.py:1233-1267,1310-1453,1577-1704
# --------------------------------------------------------------------------

class _SafeSqrt(torch.autograd.Function):
    """sqrt with gradient 0 at 0.  The reference's tf.sqrt yields NaN there
    (0*inf) and hides it with np.nan_to_num on the updated variable
    (styler_3p.py:337,340,360); the build uses the analytic limit instead
    (documented deviation)."""

    @staticmethod
    def forward(ctx, x):
        y = torch.sqrt(x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return torch.where(y > 0, g / (2 * y), torch.zeros_like(g))


def cubic_W(q, h, is_3d):
    """cubic spline (transform.py:1234-1245)."""
    sigma = 8 / math.pi / h ** 3 if is_3d else 40 / 7 / math.pi / h ** 2
    inner = torch.where(q <= 0.5, 6 * (q ** 3 - q ** 2) + 1, 2 * (1 - q) ** 3)
    return torch.where(q > 1, torch.zeros_like(q), sigma * inner)


def _splat_common(p, domain, res, nsize, clip, eps):
    """Shared prologue of p2g / p2g_wavg (transform.py:1315-1343)."""
    dt = p.dtype
    dom = torch.tensor([float(v) for v in domain], dtype=dt)
    p = p * dom
    if clip:
        p = torch.minimum(torch.clamp_min(p, 0), dom - eps)
        valid = torch.ones(p.shape[:-1], dtype=torch.bool)
    else:
        valid = ((p >= 0) & (p < dom)).all(dim=-1)
    # cell_size = (domain / res)[0] in the graph dtype (1328-1330)
    cell = float((dom[0] / torch.tensor(float(res[0]), dtype=dt)).item())
    fl = torch.floor(p / cell)
    idx = fl.long()
    r = p - (fl + 0.5) * cell
    return r, idx, valid, cell


def _scatter_nd(idx_list, upd, res, valid, C):
    """tf.scatter_nd into zeros(res+[C]); out-of-range indices are dropped as
    the TF GPU kernel does (the reference ran on GPU)."""
    ok = valid.clone()
    lin = torch.zeros_like(idx_list[0])
    for k, n in zip(idx_list, res):
        ok = ok & (k >= 0) & (k < n)
        lin = lin * n + k.clamp(0, n - 1)
    out = torch.zeros(int(np.prod(res)), C, dtype=upd.dtype)
    out = out.index_add(0, lin[ok], upd[ok])
    return out.reshape(*res, C)


def p2g(p, domain, res, radius, rest_density, nsize, pc=None, pd=None, is_2d=True,
        clip=True, support=4, eps=1e-6):
    """p [1,N,d] in [0,1] ordered (z,y,x)/(y,x) -> [1,*res,1 or C].
    transform.py:1310-1453: mass = 0.8(2r)^d rho0 (1348-1352); for every offset
    n in [-nsize,nsize]^d: q = |r - n*cell| / (radius*support) (1363,1416),
    scatter mass*W(q) at idx+n (1366-1380, 1419-1430); colour mode
    mass*W*pc/pd (1382-1390, 1433-1441); flip H at the end (1404, 1452)."""
    assert p.shape[0] == 1
    nd = 2 if is_2d else 3
    res = [int(v) for v in res]
    r, idx, valid, cell = _splat_common(p[0], domain, res, nsize, clip, eps)
    h = radius * support
    mass = 0.8 * (2 * radius) ** nd * rest_density
    C = 1 if pc is None else pc.shape[-1]
    out = 0
    offs = range(-nsize, nsize + 1)
    import itertools
    for n in itertools.product(offs, repeat=nd):
        nn = torch.tensor(n, dtype=r.dtype)
        dist = _SafeSqrt.apply(((r - nn * cell) ** 2).sum(-1))
        w = cubic_W(dist / h, h, not is_2d)
        if pc is None:
            upd = (mass * w).unsqueeze(-1)
        else:
            upd = mass * w.unsqueeze(-1) * pc[0]
            upd = upd / (rest_density if pd is None else pd[0])
        out = out + _scatter_nd([idx[:, k] + n[k] for k in range(nd)], upd, res, valid, C)
    out = out.flip(0 if is_2d else 1)      # H axis of [H,W,C] / [D,H,W,C]
    return out.unsqueeze(0)


def p2g_wavg(p, x, domain, res, radius, nsize, is_2d=True, clip=True, support=4, eps=1e-6):
    """weighted average splat (transform.py:1577-1704) with the cubic kernel as
    styler_3p.py:83 calls it: sum(w*x)/sum(w) where sum(w) > eps else sum(w*x)."""
    assert p.shape[0] == 1
    nd = 2 if is_2d else 3
    res = [int(v) for v in res]
    r, idx, valid, cell = _splat_common(p[0], domain, res, nsize, clip, eps)
    h = radius * support
    C = x.shape[-1]
    wsum = 0; xsum = 0
    import itertools
    for n in itertools.product(range(-nsize, nsize + 1), repeat=nd):
        nn = torch.tensor(n, dtype=r.dtype)
        dist = _SafeSqrt.apply(((r - nn * cell) ** 2).sum(-1))
        w = cubic_W(dist / h, h, not is_2d).unsqueeze(-1)
        ids = [idx[:, k] + n[k] for k in range(nd)]
        wsum = wsum + _scatter_nd(ids, w, res, valid, 1)
        xsum = xsum + _scatter_nd(ids, w * x[0], res, valid, C)
    ax = 0 if is_2d else 1
    wsum = wsum.flip(ax); xsum = xsum.flip(ax)
    safe = torch.where(wsum > eps, wsum, torch.ones_like(wsum))
    out = torch.where(wsum > eps, xsum / safe, xsum)
    return out.unsqueeze(0)


def p2g_numpy_loops(p, domain, res, radius, rest_density, nsize, support=4, is_2d=False):
    """Independent pure-Python restatement of p2g (density mode, clip=False) for
    tiny cases; pins the vectorised version above."""
    p = np.asarray(p, np.float64)
    nd = p.shape[-1]
    res = [int(v) for v in res]
    out = np.zeros(res, np.float64)
    cell = float(domain[0]) / res[0]
    h = radius * support
    sigma = 8 / math.pi / h ** 3 if nd == 3 else 40 / 7 / math.pi / h ** 2
    mass = 0.8 * (2 * radius) ** nd * rest_density
    import itertools
    for a in range(p.shape[0]):
        pos = p[a] * np.asarray(domain, np.float64)
        if np.any(pos < 0) or np.any(pos >= np.asarray(domain)):
            continue
        ci = np.floor(pos / cell).astype(int)
        for n in itertools.product(range(-nsize, nsize + 1), repeat=nd):
            cj = ci + np.array(n)
            if np.any(cj < 0) or np.any(cj >= np.array(res)):
                continue
            q = np.linalg.norm(pos - (cj + 0.5) * cell) / h
            if q > 1:
                continue
            w = 6 * (q ** 3 - q ** 2) + 1 if q <= 0.5 else 2 * (1 - q) ** 3
            out[tuple(cj)] += mass * sigma * w
    return np.flip(out, axis=0 if nd == 2 else 1)[None, ..., None]


# --------------------------------------------------------------------------
# A10  TF ApplyAdam
# --------------------------------------------------------------------------

class TFAdam:
    """tf.compat.v1.train.AdamOptimizer semantics (styler_3p.py:320):
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); x -= lr_t*m/(sqrt(v)+eps) -- epsilon is
    added to the un-bias-corrected sqrt(v) (differs from torch.optim.Adam)."""

    def __init__(self, beta1=0.9, beta2=0.999, eps=1e-8):
        self.b1, self.b2, self.eps = beta1, beta2, eps
        self.m = None; self.v = None; self.t = 0

    def step(self, x, g, lr):
        if self.m is None:
            self.m = torch.zeros_like(x); self.v = torch.zeros_like(x)
        self.t += 1
        self.m = self.b1 * self.m + (1 - self.b1) * g
        self.v = self.b2 * self.v + (1 - self.b2) * g * g
        # TF computes lr_t in the variable's dtype (float32 graph)
        if x.dtype == torch.float32:
            b1p = np.float32(self.b1) ** np.float32(self.t)
            b2p = np.float32(self.b2) ** np.float32(self.t)
            lr_t = float(np.float32(lr) * np.sqrt(np.float32(1) - b2p) / (np.float32(1) - b1p))
        else:
            lr_t = lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        return x - lr_t * self.m / (torch.sqrt(self.v) + self.eps)


# --------------------------------------------------------------------------
# Assembled forward graphs
# --------------------------------------------------------------------------

def grid_forward(d0, vel, rot, cfg, weights, style_feats, var="vel"):
    """TNST-style grid path assembled from the reference's operators
    (SURVEY.md section 0.1): d^ = advect(d0, vel) (transform.py:557-569) ->
    smooth+max (styler_3p.py:112-125) -> rotate (133) -> render (147-158) ->
    loss net (styler_base.py:33-57) -> style loss (152-185).
    ``rot`` [V,3,3]; views are summed (views=sum mode): each view is rendered
    and normalised separately (v_batch=1 semantics) and the losses are added.
    Returns (total_loss, per_view_losses, d_out)."""
    d = advect(d0, vel) if vel is not None else d0
    d_out = smooth3d_relu(d, cfg["k"])
    total = 0
    per_view = []
    for v in range(rot.shape[0]):
        l = grid_view_loss(d_out, rot[v:v + 1], cfg, weights, style_feats)
        per_view.append(l)
        total = total + l
    return total, per_view, d_out


def grid_view_loss(d_out, rot_v, cfg, weights, style_feats):
    """the loss of ONE view of ``grid_forward`` from the smoothed density ``d_out`` [1,D,H,W,1]: rotate
    (styler_3p.py:133) -> render (147-158) -> loss net (styler_base.py:33-57) -> style (152-185) / content /
    histogram terms; ``rot_v`` [1,3,3]"""
    dr = rotate(d_out, rot_v) if cfg.get("rotate", True) else d_out
    img = render(dr, cfg["transmit"], cfg.get("ray_mode") or cfg.get("render_liquid", False))
    d_img = plugin_to_loss_net(img, cfg.get("resize_scale", 1.0))
    feats = loss_net_features(d_img, weights, cfg["style_layer"][-1] if cfg.get("upto") is None else cfg["upto"], cfg)
    l, _ = style_loss(feats, style_feats, cfg["style_layer"], cfg["w_style_layer"], cfg.get("w_style", 1.0))
    if cfg.get("w_content", 0):
        # one view per loss-net batch here (v_batch = 1): the content means are per view
        l = l + cfg["w_content"] * content_loss(feats[cfg["content_layer"]], cfg.get("content_channel", 0),
                                                cfg.get("content_feature"), cfg.get("w_content_amp", 100.0))
    if cfg.get("w_hist", 0):
        # histogram term (styler_base.py:187-209): 'input' = d_img, otherwise a layer of the loss network
        for name, wl in zip(cfg["hist_layer"], cfg["w_hist_layer"]):
            f = d_img if "input" in name else feats[name]
            l = l + cfg["w_hist"] * wl * hist_loss(f, cfg["hist_feature"][name])
    return l


def style_target_features(style_img, weights, layers, upto=None, cfg=None):
    """_style_feature (styler_base.py:249-278): the style image is fed directly
    at d_img (0..255, before mean subtraction)."""
    feats = loss_net_features(style_img, weights, layers[-1] if upto is None else upto, cfg)
    return {k: feats[k].detach() for k in layers}


def last_layer(layers, cfg=None):
    if cfg is not None and "inception" in str(cfg.get("network", "vgg")):
        return inception_last_layer(layers)
    order = vgg19_layer_names("conv5_4")
    return max(layers, key=order.index)


def uses_content(cfg):
    """the content term is on when its layer is a tensor of the chosen network (the config default names an Inception
    tensor: with the VGG network the drivers of this build switch the term off)"""
    if not cfg.get("w_content", 0):
        return False
    if "inception" in str(cfg.get("network", "vgg")):
        return True
    return str(cfg.get("content_layer", "")).startswith("conv")


# --------------------------------------------------------------------------
# A10  outer loops (styler_3p.py:229-439, styler_2p.py:165-315) -- CPU restatement
# --------------------------------------------------------------------------

def particle_field(p, r, var, cfg, res):
    """styler_3p.py:42-128: variable -> (positions, d_out [1,D,H,W,1], auxiliary loss or None).
    Auxiliary terms: pressure loss of the 'p' field (styler_3p.py:96-98, styler_base.py:226-230) and the
    density-preservation loss on the clipped density offsets ``self.d[i]`` of the 'd' field
    (styler_3p.py:75, styler_base.py:217-223)."""
    tf_ = cfg["target_field"]
    p_ = p.unsqueeze(0)
    extra = None
    if "p" in tf_:
        p_ = p_ + var.unsqueeze(0)
    if "d" in tf_:
        r_opt = torch.clamp(var.unsqueeze(0), -1, 1)
        r_ = r.unsqueeze(0) + r_opt
        if cfg.get("w_density", 0) > 0:
            d_loss = r_opt[0].sum() ** 2
            d_pres = (-torch.log(r_opt[0].abs() + 1e-6)).sum()
            extra = (d_loss + d_pres * 1e3) * cfg["w_density"]
        d_ = 0
        for k in range(cfg["num_kernels"]):
            support = cfg["support"] / cfg["kernel_scale"] ** k
            d_ = d_ + p2g_wavg(p_, r_[..., k:k + 1], cfg["domain"], res, cfg["radius"], cfg["nsize"], is_2d=False,
                               clip=cfg["clip"], support=support)
    else:
        d_ = p2g(p_, cfg["domain"], res, cfg["radius"], cfg["rest_density"], cfg["nsize"], is_2d=False,
                 clip=cfg["clip"], support=cfg["support"]) / cfg["rest_density"]
        if cfg.get("w_pressure", 0) > 0:
            pressure = torch.where(d_ > 0, d_ - 1, torch.zeros_like(d_))
            extra = (pressure ** 2).mean() * cfg["w_pressure"]
    return p_[0], smooth3d_relu(d_, cfg["k"]), extra


def particle_loss(p, r, var, cfg, res, rot, weights, style_feats):
    _, d_out, extra = particle_field(p, r, var, cfg, res)
    total = 0
    for v in range(rot.shape[0]):
        dr = rotate(d_out, rot[v:v + 1]) if cfg["rotate"] else d_out
        img = render(dr, cfg["transmit"], cfg.get("ray_mode") or cfg.get("render_liquid", False))
        d_img = plugin_to_loss_net(img, cfg.get("resize_scale", 1.0))
        use_content = uses_content(cfg)
        feats = loss_net_features(d_img, weights, last_layer(cfg["style_layer"] +
                                                             ([cfg["content_layer"]] if use_content else []), cfg), cfg)
        l, _ = style_loss(feats, style_feats, cfg["style_layer"], cfg["w_style_layer"], cfg["w_style"])
        if use_content:                                           # styler_base.py:135-150, total_loss order: content first
            l = l + cfg["w_content"] * content_loss(feats[cfg["content_layer"]], cfg.get("content_channel", 0),
                                                    cfg.get("content_feature"), cfg.get("w_content_amp", 100.0))
        if cfg.get("w_tv", 0):
            l = l + tv_loss(d_img) * cfg["w_tv"]
        total = total + l
    if extra is not None:
        total = total + extra
    return total


def styler3p_run(cfg, params, weights, style_img, rot_mats, views_mode="sequential"):
    """The reference's Styler.run for one octave list / uniform views (no Poisson re-sampling):
    returns (loss history per octave, list of optimised variables, final d_out per frame)."""
    from scipy.ndimage import gaussian_filter
    dt = torch.float32
    F_ = cfg["num_frames"]
    p = [torch.tensor(np.asarray(x), dtype=dt) for x in params["p"]]
    r = [torch.tensor(np.asarray(x), dtype=dt) for x in params["r"]] if "d" in cfg["target_field"] else [None] * F_
    nvar = 3 if "p" in cfg["target_field"] else cfg["num_kernels"]
    g_opt = [torch.zeros(p[i].shape[0], nvar, dtype=dt) for i in range(F_)]
    oct_size = []
    dhw = np.array(cfg["resolution"])
    for _ in range(cfg["octave_n"]):
        oct_size.append(dhw)
        dhw = (dhw // cfg["octave_scale"]).astype(int)
    oct_size.reverse()
    rot_all = torch.tensor(np.asarray(rot_mats, np.float32)) if cfg["rotate"] else torch.eye(3)[None]
    opt_ = {}
    hist = []
    for octave in range(cfg["octave_n"]):
        res = [int(v) for v in oct_size[octave]]
        simg = torch.tensor(np.asarray(style_img[octave], np.float32))[None]
        sfe = style_target_features(simg, weights, cfg["style_layer"], upto=last_layer(cfg["style_layer"], cfg), cfg=cfg)
        h_o = []
        for step in range(cfg["iter"]):
            g_tmp = [None] * F_
            for t in range(0, F_, cfg["interp"]):
                var = g_opt[t].clone()
                opt = opt_.setdefault(t // cfg["frames_per_opt"], TFAdam())

                def grad_at(v, rot):
                    vv = v.clone().requires_grad_()
                    l = particle_loss(p[t], r[t], vv, cfg, res, rot, weights, sfe)
                    (g,) = torch.autograd.grad(l, vv)
                    return l.detach(), g

                if cfg["rotate"] and views_mode == "sequential":
                    acc, ls = None, []
                    vb = cfg["v_batch"]
                    for i in range(0, rot_all.shape[0], vb):
                        l, g = grad_at(var, rot_all[i:i + vb])
                        var = opt.step(var, g, cfg["lr"])
                        ls.append(float(l))
                        cur = torch.nan_to_num(var)
                        acc = cur.clone() if acc is None else acc + cur
                    h_o.append(float(np.mean(ls)))
                    new = acc / (rot_all.shape[0] / vb)
                else:
                    l, g = grad_at(var, rot_all)
                    var = opt.step(var, g, cfg["lr"])
                    h_o.append(float(l.detach()))
                    new = torch.nan_to_num(var)
                upd = new - g_opt[t]
                if "d" in cfg["target_field"]:
                    upd = upd * r[t][..., 0:1]
                g_tmp[t] = upd
            idx = list(range(0, F_, cfg["interp"]))
            if cfg["window_sigma"] > 0 and F_ > 1:
                st = gaussian_filter(np.stack([g_tmp[i].numpy() for i in idx]), sigma=(cfg["window_sigma"], 0, 0))
                for j, i in enumerate(idx):
                    g_tmp[i] = torch.tensor(st[j])
            for i in idx:
                g_opt[i] = g_opt[i] + g_tmp[i]
        hist.append(h_o)
    # frame interpolation (styler_3p.py:392-397): the frames between two key frames get the linear blend of their
    # variables (the reference indexes g_opt[t + interp] unguarded: num_frames - 1 is a multiple of interp in its runs)
    if cfg.get("interp", 1) > 1:
        w_ = np.linspace(0, 1, cfg["interp"] + 1)
        for t in range(0, F_ - 1, cfg["interp"]):
            for i in range(1, cfg["interp"]):
                if t + cfg["interp"] < F_:
                    g_opt[t + i] = g_opt[t] * float(1 - w_[i]) + g_opt[t + cfg["interp"]] * float(w_[i])
    res = [int(v) for v in oct_size[-1]]
    d_fin = [particle_field(p[t], r[t], g_opt[t], cfg, res)[1][0] for t in range(F_)]
    return hist, g_opt, d_fin


def temporal_filter_matrix(n, sigma):
    """The frame-axis filter of ``denoise`` (util.py:169-170) as a matrix, obtained by pushing the identity through
    SciPy's own gaussian_filter (defaults: reflect, truncate 4 sigma): column s = response to a unit update of frame s."""
    from scipy.ndimage import gaussian_filter
    if sigma <= 0 or n <= 1:
        return np.eye(n)
    return gaussian_filter(np.eye(n), sigma=(sigma, 0))


def grid_sequence_run(cfg, d_frames, u_frames, weights, style_img, rot_mats, v_init=None):
    """Grid counterpart of Styler.run (styler_3p.py:300-397) assembled from the reference's grid leftovers -- the
    restatement the HIP grid-sequence stylizer (styler_grid.py) is checked against.

    Per key frame t (304): variable re-assigned from g_opt[t] (312), TFAdam per ``t // frames_per_opt`` (315-323), one
    step on the summed view losses of grid_forward (advect 557-569 -> smooth 112-125 -> rotate 611-628 -> render
    147-158 -> style loss); update = new - old (359-360), 'd' masked by the original density (361-363).  Gradient
    alignment (380-386): the per-frame updates are Gaussian-filtered along the frame axis; a grid field is first
    carried to the receiving frame by ``transport`` (= StylerBase._transport, styler_base.py:59-89), each source frame
    separately (no Horner form here).  Then g_opt[t] += (386) and the frame interpolation (392-397).
    d_frames [F,D,H,W], u_frames [F,D,H,W,3] (advect units).  Returns (loss history [iter][key frame], list of
    variables per frame [D,H,W,C], list of final d_out per frame)."""
    F_ = cfg["num_frames"]
    interp = cfg.get("interp", 1)
    target = cfg.get("grid_variable", "v")
    dt = torch.float32
    d = [torch.tensor(np.asarray(x), dtype=dt)[None, ..., None] for x in d_frames]
    u = torch.tensor(np.asarray(u_frames), dtype=dt) if u_frames is not None else None
    rot = torch.tensor(np.asarray(rot_mats, np.float32)) if cfg.get("rotate", True) else torch.eye(3)[None]
    sfe = style_target_features(torch.tensor(np.asarray(style_img, np.float32))[None], weights, cfg["style_layer"],
                                upto=cfg.get("upto"), cfg=cfg)
    keys = list(range(0, F_, interp))
    g_opt = {}
    for t in keys:
        if target == "d":
            g_opt[t] = d[t].clone()
        elif v_init is not None:
            g_opt[t] = torch.tensor(np.asarray(v_init[t]), dtype=dt)[None]
        else:
            g_opt[t] = torch.zeros(d[t].shape[:-1] + (3,), dtype=dt)
    Wm = temporal_filter_matrix(len(keys), cfg["window_sigma"]) if (cfg["window_sigma"] > 0 and F_ > 1) else None
    opt_ = {}
    hist = []
    for step in range(cfg["iter"]):
        upd, h = {}, []
        for t in keys:
            var = g_opt[t].clone().requires_grad_()
            opt = opt_.setdefault(t // cfg["frames_per_opt"], TFAdam())
            if target == "v":
                total, _, _ = grid_forward(d[t], var, rot, cfg, weights, sfe)
            else:
                total, _, _ = grid_forward(var, None, rot, cfg, weights, sfe)
            (g,) = torch.autograd.grad(total, var)
            new = torch.nan_to_num(opt.step(var.detach(), g, cfg["lr"]))
            dl = new - g_opt[t]
            if target == "d":
                dl = dl * d[t]
            upd[t] = dl
            h.append(float(total))
        hist.append(h)
        for j, t in enumerate(keys):
            if Wm is None:
                g_opt[t] = g_opt[t] + upd[t]
                continue
            acc = torch.zeros_like(upd[t])
            for jj, s in enumerate(keys):
                if Wm[j, jj] == 0.0:
                    continue
                acc = acc + float(Wm[j, jj]) * transport(upd[s], u, s, t, recursive=cfg.get("transport_recursive", True))
            g_opt[t] = g_opt[t] + acc
    full = dict(g_opt)
    if interp > 1:
        w = np.linspace(0, 1, interp + 1)
        for t in range(0, F_ - 1, interp):
            for i in range(1, interp):
                if t + interp < F_:
                    full[t + i] = full[t] * float(1 - w[i]) + full[t + interp] * float(w[i])
    outs, d_fin = [], []
    for t in range(F_):
        var = full.get(t)
        if var is None:
            if target == "d":
                var = d[t].clone()
            elif v_init is not None:
                var = torch.tensor(np.asarray(v_init[t]), dtype=dt)[None]
            else:
                var = torch.zeros(d[t].shape[:-1] + (3,), dtype=dt)
        d_adv = advect(d[t], var) if target == "v" else var
        d_fin.append(smooth3d_relu(d_adv, cfg["k"])[0])
        outs.append(var[0])
    return hist, outs, d_fin


def colour_field2d(p, r, var, cfg, res):
    """styler_2p.py:42-102: d_gray = clip(p2g(p)/rho0, 0, 1) (mask, constant); d = clip(p2g(p, pc=clip(c,0,1),
    pd=r), 0, 1) [1,H,W,3]; returns (d, d_gray, clipped colours)"""
    pb = p.unsqueeze(0)
    d_gray = torch.clamp(p2g(pb, cfg["domain"], res, cfg["radius"], cfg["rest_density"], cfg["nsize"], is_2d=True,
                             clip=cfg["clip"], support=cfg["support"]) / cfg["rest_density"], 0, 1)
    c_ = torch.clamp(var.unsqueeze(0), 0, 1)
    d = p2g(pb, cfg["domain"], res, cfg["radius"], cfg["rest_density"], cfg["nsize"], pc=c_, pd=r.unsqueeze(0),
            is_2d=True, clip=cfg["clip"], support=cfg["support"])
    return torch.clamp(d, 0, 1), d_gray.detach(), c_[0]


def colour_loss2d(p, r, var, cfg, res, weights, style_feats, batch=1):
    """style (optionally masked by d_gray, styler_base.py:165-169) + TV (211-213) of the colour image.  ``batch``: the
    image is one of ``batch`` frames of a sess.run (styler_2p.py:42-98 builds batch_size towers): the style term is a
    reduce_sum over the batch (181: separable), the TV term a reduce_mean (212: this frame's share is TV / batch)"""
    d, d_gray, _ = colour_field2d(p, r, var, cfg, res)
    d_img = plugin_to_loss_net(d, cfg.get("resize_scale", 1.0), is_color=True)
    use_content = uses_content(cfg)
    vgg_hist = [n for n in cfg.get("hist_layer", []) if "input" not in n] if cfg.get("w_hist", 0) else []
    feats = loss_net_features(d_img, weights, last_layer(cfg["style_layer"] + vgg_hist +
                                                         ([cfg["content_layer"]] if use_content else []), cfg), cfg)
    l, _ = style_loss(feats, style_feats, cfg["style_layer"], cfg["w_style_layer"], cfg["w_style"],
                      d_gray=d_gray if cfg.get("style_mask") else None)
    if use_content:
        # (every form of the content term is a reduce_mean over the batch tensor, styler_base.py:135-150: 1 / batch)
        l = l + cfg["w_content"] * content_loss(feats[cfg["content_layer"]], cfg.get("content_channel", 0),
                                                cfg.get("content_feature"), cfg.get("w_content_amp", 100.0)) / batch
    if cfg.get("w_hist", 0):
        # histogram term (styler_base.py:187-209); with style_mask the masked branch (196-201): the density mask,
        # bicubic-resized to the layer, removes its zero pixels from the source of the match
        for name, wl in zip(cfg["hist_layer"], cfg["w_hist_layer"]):
            f = d_img if "input" in name else feats[name]
            m = tf1_resize_bicubic(d_gray, f.shape[1], f.shape[2]) if cfg.get("style_mask") else None
            l = l + cfg["w_hist"] * wl * hist_loss(f, cfg["hist_feature"][name], mask=m)
    if cfg.get("w_tv", 0):
        l = l + tv_loss(d_img) * cfg["w_tv"] / batch
    return l


def styler2p_run(cfg, params, weights, style_img, c_init):
    """The reference's 2-D colour Styler.run (styler_2p.py:165-315): octaves coarse -> fine, one TF-Adam state per
    frame group, temporal Gaussian smoothing of the per-frame updates.  ``batch_size`` B > 1 (run.bat's last line): B
    consecutive frames share one sess.run -- total loss = sum of their style terms + mean of their TV terms, ONE
    optimiser step on the B colour variables (243-255: the Adam slots belong to the batch POSITION, the step count to
    the optimiser), one loss entry per batch (258).  Style, content (a mean over the batch) and TV there; the unmasked
    histogram match runs over the whole batch tensor and is not restated for B > 1.  ``style_img[octave]`` is
    the style image already resized for that octave, ``c_init`` [F,N,3] the colour initialisation (189-192).
    Returns (loss history per octave, optimised colours per frame, final uint8 images d*d_gray*255)."""
    from scipy.ndimage import gaussian_filter
    dt = torch.float32
    F_ = cfg["num_frames"]
    p = [torch.tensor(np.asarray(x), dtype=dt) for x in params["p"]]
    r = [torch.tensor(np.asarray(x), dtype=dt) for x in params["r"]]
    g_opt = [torch.tensor(np.asarray(c_init[i]), dtype=dt) for i in range(F_)]
    oct_size = []
    hw = np.array(cfg["resolution"])
    for _ in range(cfg["octave_n"]):
        oct_size.append(hw)
        hw = (hw // cfg["octave_scale"]).astype(int)
    oct_size.reverse()
    opt_ = {}
    hist = []
    for octave in range(cfg["octave_n"]):
        res = [int(v) for v in oct_size[octave]]
        simg = torch.tensor(np.asarray(style_img[octave], np.float32))[None]
        sfe = style_target_features(simg, weights, cfg["style_layer"], upto=last_layer(cfg["style_layer"], cfg), cfg=cfg)
        lr = cfg["lr"][octave] if isinstance(cfg["lr"], (list, tuple)) else cfg["lr"]
        h_o = []
        for step in range(cfg["iter"]):
            g_tmp = [None] * F_
            B = int(cfg.get("batch_size", 1) or 1)
            assert F_ % B == 0, "num_frames must be a multiple of batch_size (styler_2p.py:239-244 indexes p[t+i])"
            assert B == 1 or not cfg.get("w_hist", 0), "batch_size > 1: no histogram term (it matches over the batch tensor)"
            for t in range(0, F_, B):
                opt = opt_.setdefault(t // cfg["frames_per_opt"], TFAdam())
                vv = [g_opt[t + i].clone().requires_grad_() for i in range(B)]
                l = sum(colour_loss2d(p[t + i], r[t + i], vv[i], cfg, res, weights, sfe, batch=B) for i in range(B))
                g = torch.autograd.grad(l, vv)
                h_o.append(float(l.detach()))
                new = torch.nan_to_num(opt.step(torch.stack([g_opt[t + i] for i in range(B)]), torch.stack(g), lr))
                for i in range(B):
                    g_tmp[t + i] = new[i] - g_opt[t + i]
            if cfg["window_sigma"] > 0 and F_ > 1:
                st = gaussian_filter(np.stack([g.numpy() for g in g_tmp]), sigma=(cfg["window_sigma"], 0, 0))
                g_tmp = [torch.tensor(s) for s in st]
            for t in range(F_):
                g_opt[t] = g_opt[t] + g_tmp[t]
        hist.append(h_o)
    res = [int(v) for v in oct_size[-1]]
    imgs = []
    for t in range(F_):
        d, d_gray, _ = colour_field2d(p[t], r[t], g_opt[t], cfg, res)
        imgs.append(((d * d_gray)[0] * 255).numpy().astype(np.uint8))
    return hist, g_opt, imgs


# --------------------------------------------------------------------------
# grid -> particle sampling and the SimG2P resampler
# (transform.py:771-1231, test_smokegun_resim.py:17-217).  Parity unpinned (the reference holds no
# vectors for these); pinned here by analytic known answers (tests/test_oracle_kat.py).
# --------------------------------------------------------------------------
def _g2p_axis(x, n, cubic):
    """one axis of g2p: cell-centred coordinate x in [0,n] -> (clipped indices, weights).
    linear (transform.py:1140-1154, 1196-1197): x0 = floor(x-0.5), x1 = x0+1, both clipped to [0,n-1],
    dx = x - (clipped x0 + 0.5), weights (1-dx, dx).
    cubic (transform.py:811-836, 972-978, 1000-1001): x1 = floor(x-0.5); x0..x3 = x1-1..x1+2 clipped;
    t = x - (clipped x1 + 0.5); Catmull-Rom weights of _hermite."""
    i = torch.floor(x - 0.5).long()
    if not cubic:
        i0 = i.clamp(0, n - 1); i1 = (i + 1).clamp(0, n - 1)
        dx = x - (i0.to(x.dtype) + 0.5)
        return [i0, i1], [1.0 - dx, dx]
    idx = [(i + k).clamp(0, n - 1) for k in (-1, 0, 1, 2)]
    t = x - (idx[1].to(x.dtype) + 0.5)
    t2, t3 = t * t, t * t * t
    w = [-0.5 * t3 + t2 - 0.5 * t, 1.5 * t3 - 2.5 * t2 + 1.0, -1.5 * t3 + 2.0 * t2 + 0.5 * t, 0.5 * t3 - 0.5 * t2]
    return idx, w


def g2p(g, p, is_2d=True, is_linear=False):
    """g [1,X,Y,(Z),C], p [1,N,d] in [0,1] (axis order = array order) -> [1,N,C]
    (transform.py:771-776: cubic unless is_linear)."""
    assert g.shape[0] == 1 and p.shape[0] == 1
    nd = 2 if is_2d else 3
    dims = list(g.shape[1:1 + nd])
    C = g.shape[-1]
    gf = g.reshape(-1, C)
    ax = [_g2p_axis(p[0, :, a] * dims[a], dims[a], not is_linear) for a in range(nd)]
    out = 0
    import itertools
    for combo in itertools.product(*[range(len(ax[a][0])) for a in range(nd)]):
        flat = 0
        w = 1.0
        for a in range(nd):
            flat = flat * dims[a] + ax[a][0][combo[a]]
            w = w * ax[a][1][combo[a]]
        out = out + w.unsqueeze(-1) * gf[flat]
    return out.unsqueeze(0)


def mac_to_centered(v_):
    """mantaflow MAC-grid velocity [D,H,W,3] (x,y,z components on the low faces) -> cell-centred,
    H flipped (test_smokegun_resim.py:233-243), numpy."""
    vx = np.dstack((v_, np.zeros((v_.shape[0], v_.shape[1], 1, v_.shape[3]), v_.dtype)))
    vx = (vx[:, :, 1:, 0] + vx[:, :, :-1, 0]) * 0.5
    vy = np.hstack((v_, np.zeros((v_.shape[0], 1, v_.shape[2], v_.shape[3]), v_.dtype)))
    vy = (vy[:, 1:, :, 1] + vy[:, :-1, :, 1]) * 0.5
    vz = np.vstack((v_, np.zeros((1, v_.shape[1], v_.shape[2], v_.shape[3]), v_.dtype)))
    vz = (vz[1:, :, :, 2] + vz[:-1, :, :, 2]) * 0.5
    v = np.stack([vx, vy, vz], axis=-1)
    return v[:, ::-1]


def simg2p_advect(x, u):
    """RK4 velocity sampling + advection with time step 0.5 (test_smokegun_resim.py:35-53).
    x [N,3] in [0,1] (z,y,x), u [D,H,W,3] -> x_adv [N,3]"""
    xb, ub = x.unsqueeze(0), u.unsqueeze(0)
    v = g2p(ub, xb, is_2d=False)
    v1 = g2p(ub, xb + v * 0.5, is_2d=False)
    v2 = g2p(ub, xb + v1 * 0.5, is_2d=False)
    v3 = g2p(ub, xb + v2, is_2d=False)
    v = (v + v1 * 2 + v2 * 2 + v3) / 6
    return (xb + v * 0.5)[0]


def simg2p_pressure_loss(x_hat, cfg, res):
    """mean(pressure^2), pressure = where(d_rec > 0, d_rec - rho0, 0) of the cubic splat with support 4,
    clip=False (test_smokegun_resim.py:65-71)"""
    d_rec = p2g(x_hat.unsqueeze(0), cfg["domain"], res, cfg["radius"], cfg["rest_density"], cfg["nsize"],
                is_2d=False, clip=False, support=4)
    pres = torch.where(d_rec > 0, d_rec - cfg["rest_density"], torch.zeros_like(d_rec))
    return (pres ** 2).mean()


def simg2p_density_sampling(x_hat, d, cfg, res):
    """multi-scale particle density sampling (test_smokegun_resim.py:83-106): per octave sample the residual
    d - d_hat_prev[:, :, ::-1] at the particles (cubic g2p), splat it back with support/octave_scale^o.
    Returns r_smp [N,octave_n], d_smp [D,H,W] (clipped to 0..1), d_diff [D,H,W]"""
    dd = d.unsqueeze(0).unsqueeze(-1)
    xb = x_hat.unsqueeze(0)
    r = []
    d_hat = None
    for o in range(cfg["octave_n"]):
        if o > 0:
            d_hi = d_hat
            d_ = dd - d_hi.flip(2)
        else:
            d_ = dd
        r_ = g2p(d_, xb, is_2d=False)
        r.append(r_)
        factor = cfg["octave_scale"] ** o
        d_hat = p2g_wavg(xb, r_, cfg["domain"], res, cfg["radius"], cfg["nsize"], is_2d=False, clip=False,
                         support=cfg["support"] / factor)
        if o > 0:
            d_hat = d_hat + d_hi
    r_smp = torch.cat(r, dim=-1)[0]
    d_smp = d_hat[0, ..., 0].clamp(0, 1)
    d_diff = (dd.flip(2) - d_hat)[0].flip(1)[..., 0]
    return r_smp, d_smp, d_diff


def simg2p_optimize(p, d, u, cfg, res):
    """SimG2P.optimize up to (not including) the seeding of new particles (test_smokegun_resim.py:168-199):
    advect, `iter` TF-Adam steps on the particle displacement against the pressure loss, then the density
    sampling at the redistributed positions.  Returns dict(p_adv, p_new, l, d_diff, r_smp, d_smp)."""
    p_adv = simg2p_advect(p, u)
    v = torch.zeros_like(p_adv)
    opt = TFAdam()
    losses = []
    for _ in range(cfg["iter"]):
        vv = v.clone().requires_grad_()
        loss = simg2p_pressure_loss(p_adv + vv, cfg, res)
        (g,) = torch.autograd.grad(loss, vv)
        losses.append(float(loss))
        v = opt.step(v, g, cfg["lr"])
    p_new = p_adv + v
    r_smp, d_smp, d_diff = simg2p_density_sampling(p_new, d, cfg, res)
    return dict(p_adv=p_adv, p_new=p_new, l=losses, d_diff=d_diff, r_smp=r_smp, d_smp=d_smp)
