"""Tensor-level bindings of the C ABI (include/nfs_hip.h).

PyTorch is plumbing only: it owns device memory and the stream.  Every function
takes contiguous float32 CUDA tensors, passes raw pointers + the current stream
through ctypes and returns freshly allocated (or caller-supplied) outputs.
There is no CPU path: a non-CUDA tensor raises.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import SplatCfg


def _ptr(t):
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError("expected a contiguous float32 CUDA tensor, got %s %s contiguous=%s"
                         % (t.device, t.dtype, t.is_contiguous()))
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device=None):
    """the current HIP stream's handle (the raw getter is ~20x cheaper than building a torch.cuda.Stream object: the
    one-view and small-grid steps are bound by what the host needs per launch)"""
    if _raw_stream is not None:
        idx = torch.cuda.current_device() if device is None or device.index is None else device.index
        return _raw_stream(idx)
    return torch.cuda.current_stream(device).cuda_stream


def _empty(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def _zeros(shape, like):
    return torch.zeros(shape, dtype=torch.float32, device=like.device)


# ---- A2 / A3 / A11 ------------------------------------------------------------------

def warp3d_fwd(imgs, coords):
    B, X, Y, Z, Cn = imgs.shape
    out = _empty(imgs.shape, imgs)
    _lib.call("nfs_warp3d_fwd", _ptr(imgs), _ptr(coords), _ptr(out), B, X, Y, Z, Cn, _stream())
    return out


def warp3d_bwd(imgs, coords, g_out, need_coords=True):
    B, X, Y, Z, Cn = imgs.shape
    g_imgs = _zeros(imgs.shape, imgs)
    g_coords = _empty(coords.shape, imgs) if need_coords else None
    _lib.call("nfs_warp3d_bwd", _ptr(imgs), _ptr(coords), _ptr(g_out), _ptr(g_imgs), _ptr(g_coords),
              B, X, Y, Z, Cn, _stream())
    return g_imgs, g_coords


def rotate_fwd(d, rot):
    """d [D,H,W,C], rot [V,3,3] -> [V,D,H,W,C]"""
    D, H, W, Cn = d.shape
    V = rot.shape[0]
    out = _empty((V, D, H, W, Cn), d)
    _lib.call("nfs_rotate_fwd", _ptr(d), _ptr(rot), _ptr(out), V, D, H, W, Cn, _stream())
    return out


_workspaces = {}


def workspace(device):
    """small per-device scratch (64 floats) for entry points that need a device-side scalar"""
    if torch.cuda.is_current_stream_capturing():
        # a capture owns what it allocates (its private pool keeps it for the replays); a buffer cached here would be
        # shared by every later capture on the same capture stream, whatever became of the graph that allocated it
        return torch.zeros(64, dtype=torch.float32, device=device)
    key = (device.type, device.index, _stream(device))   # one per stream
    if key not in _workspaces:
        _workspaces[key] = torch.zeros(64, dtype=torch.float32, device=device)
    return _workspaces[key]


def rotate_bwd(g_out, rot, g_d_acc=None, tiled=True, g_max=None, overwrite=False):
    """g_max: optional device scalar max|g_out| (render_bwd(..., want_max=True)): skips the pre-pass.
    overwrite (tiled adjoint, C == 1): g_d_acc is written, not accumulated into (no zero fill needed)"""
    V, D, H, W, Cn = g_out.shape
    if tiled and Cn == 1:
        # the scratch word holds the pre-pass maximum: with g_max given it is not touched (inside a capture workspace() is a
        # fresh zero-filled tensor, i.e. a fill launch in every replay: not needed here)
        ws = _empty((64,), g_out) if (g_max is not None and torch.cuda.is_current_stream_capturing()) else workspace(g_out.device)
    else:
        ws = None
    if g_d_acc is None:
        overwrite = ws is not None                 # a fresh buffer: let the kernel write every voxel
        g_d_acc = _empty((D, H, W, Cn), g_out) if overwrite else _zeros((D, H, W, Cn), g_out)
    _lib.call("nfs_rotate_bwd", _ptr(g_out), _ptr(rot), _ptr(g_d_acc), V, D, H, W, Cn, _ptr(ws), _ptr(g_max),
              int(bool(overwrite)), _stream())
    return g_d_acc


def live_mask(D, H, W, like):
    """buffer for the live mask of a [D,H,W] volume: nfs_live_mask_words 64-bit words, bit = linear voxel index, held as
    float32 storage like every buffer that crosses the C-ABI here (``.view(torch.int64)`` to look at the words)"""
    return torch.zeros(2 * int(_lib.lib().nfs_live_mask_words(D, H, W)), dtype=torch.float32, device=like.device)


def advect_fwd(d, vel, out=None, live=None):
    """``live`` (optional, ``live_mask``; scalar field only): also write the mask of the voxels whose back-traced density
    corners differ, i.e. where the velocity gradient can be non-zero (see rotate_bwd_coef)"""
    D, H, W, Cn = d.shape
    if out is None:
        out = _empty(d.shape, d)
    if live is not None:
        assert Cn == 1
        _lib.call("nfs_advect_fwd_live", _ptr(d), _ptr(vel), _ptr(out), _ptr(live), D, H, W, _stream())
        _written(live)
    else:
        _lib.call("nfs_advect_fwd", _ptr(d), _ptr(vel), _ptr(out), D, H, W, Cn, _stream())
    return out


def advect_bwd(d, vel, g_out, need_d=True, need_vel=True, g_d_acc=None, g_vel=None):
    D, H, W, Cn = d.shape
    if need_d and g_d_acc is None:
        g_d_acc = _zeros(d.shape, d)
    if need_vel and g_vel is None:
        g_vel = _empty(vel.shape, d)
    _lib.call("nfs_advect_bwd", _ptr(d), _ptr(vel), _ptr(g_out), _ptr(g_d_acc if need_d else None),
              _ptr(g_vel if need_vel else None), D, H, W, Cn, _stream())
    return g_d_acc, g_vel


def advect_bwd_adam(d, vel, g_out, m, v, lr_t, beta1=0.9, beta2=0.999, eps=1e-8, adv_next=None, live_next=None, ever=None):
    """velocity gradient of advect consumed on the spot by the TF-Adam update of vel (vel, m, v in place);
    ``adv_next`` [D,H,W] (optional): advect(d, updated vel), the next iteration's forward sample, written in the same pass;
    ``live_next`` (optional, with adv_next): the live mask of that sample (``advect_fwd``);
    ``ever`` (optional, with live_next, which must then hold the CURRENT mask on entry): the OR of every mask since m and v
    were zeroed -- waves of voxels that never were live are left out altogether (ApplyAdam is an exact no-op there)"""
    D, H, W, Cn = d.shape
    assert Cn == 1
    if adv_next is not None and live_next is not None and ever is not None:
        assert adv_next.is_contiguous() and adv_next.numel() == D * H * W and ever.numel() == live_next.numel()
        _lib.call("nfs_advect_bwd_adam_fwd_live_ever", _ptr(d), _ptr(vel), _ptr(g_out), _ptr(m), _ptr(v), _ptr(adv_next),
                  _ptr(live_next), _ptr(ever), D, H, W, float(lr_t), float(beta1), float(beta2), float(eps), _stream())
        _written(live_next, ever)
    elif adv_next is not None and live_next is not None:
        assert adv_next.is_contiguous() and adv_next.numel() == D * H * W
        _lib.call("nfs_advect_bwd_adam_fwd_live", _ptr(d), _ptr(vel), _ptr(g_out), _ptr(m), _ptr(v), _ptr(adv_next),
                  _ptr(live_next), D, H, W, float(lr_t), float(beta1), float(beta2), float(eps), _stream())
        _written(live_next)
    elif adv_next is None:
        _lib.call("nfs_advect_bwd_adam", _ptr(d), _ptr(vel), _ptr(g_out), _ptr(m), _ptr(v), D, H, W, float(lr_t),
                  float(beta1), float(beta2), float(eps), _stream())
    else:
        assert adv_next.is_contiguous() and adv_next.numel() == D * H * W
        _lib.call("nfs_advect_bwd_adam_fwd", _ptr(d), _ptr(vel), _ptr(g_out), _ptr(m), _ptr(v), _ptr(adv_next), D, H, W,
                  float(lr_t), float(beta1), float(beta2), float(eps), _stream())
    _written(vel, m, v, adv_next)


def advect_fwd_slab(d, vel_slab, z0, out=None):
    """planes [z0, z0 + nz) of advect(d, vel): d the WHOLE density [D,H,W], vel_slab [nz,H,W,3] -> [nz,H,W]"""
    D, H, W = d.shape
    nz = vel_slab.shape[0]
    if out is None:
        out = _empty((nz, H, W), d)
    _lib.call("nfs_advect_fwd_slab", _ptr(d), _ptr(vel_slab), _ptr(out), D, H, W, int(z0), nz, _stream())
    return out


def advect_bwd_adam_slab(d, vel_slab, g_slab, m_slab, v_slab, z0, lr_t, beta1=0.9, beta2=0.999, eps=1e-8, adv_next=None):
    """advect_bwd_adam on the planes [z0, z0 + nz) only (vel_slab, m_slab, v_slab [nz,H,W,3] in place, g_slab [nz,H,W]);
    d is the whole density [D,H,W]; ``adv_next`` [nz,H,W] (optional) as in advect_bwd_adam"""
    D, H, W = d.shape
    nz = vel_slab.shape[0]
    if adv_next is None:
        _lib.call("nfs_advect_bwd_adam_slab", _ptr(d), _ptr(vel_slab), _ptr(g_slab), _ptr(m_slab), _ptr(v_slab), D, H, W,
                  int(z0), nz, float(lr_t), float(beta1), float(beta2), float(eps), _stream())
    else:
        assert adv_next.is_contiguous() and adv_next.numel() == nz * H * W
        _lib.call("nfs_advect_bwd_adam_fwd_slab", _ptr(d), _ptr(vel_slab), _ptr(g_slab), _ptr(m_slab), _ptr(v_slab),
                  _ptr(adv_next), D, H, W, int(z0), nz, float(lr_t), float(beta1), float(beta2), float(eps), _stream())
    _written(vel_slab, m_slab, v_slab, adv_next)


def warp2d_fwd(imgs, coords):
    """imgs [B,X,Y,C], coords [B,2,X,Y] -> [B,X,Y,C]  (transform.py:206-236, 280-341)"""
    B, X, Y, Cn = imgs.shape
    out = _empty(imgs.shape, imgs)
    _lib.call("nfs_warp2d_fwd", _ptr(imgs), _ptr(coords), _ptr(out), B, X, Y, Cn, _stream())
    return out


def warp2d_bwd(imgs, coords, g_out, need_imgs=True, need_coords=True):
    B, X, Y, Cn = imgs.shape
    g_imgs = _zeros(imgs.shape, imgs) if need_imgs else None
    g_coords = _empty(coords.shape, imgs) if need_coords else None
    _lib.call("nfs_warp2d_bwd", _ptr(imgs), _ptr(coords), _ptr(g_out), _ptr(g_imgs), _ptr(g_coords), B, X, Y, Cn,
              _stream())
    return g_imgs, g_coords


def advect2d_fwd(d, vel):
    """d [H,W,C], vel [H,W,2] -> [H,W,C]  (transform.py:583-588)"""
    H, W, Cn = d.shape
    out = _empty(d.shape, d)
    _lib.call("nfs_advect2d_fwd", _ptr(d), _ptr(vel), _ptr(out), H, W, Cn, _stream())
    return out


def advect2d_bwd(d, vel, g_out, need_d=True, need_vel=True):
    H, W, Cn = d.shape
    g_d = _zeros(d.shape, d) if need_d else None
    g_vel = _empty(vel.shape, d) if need_vel else None
    _lib.call("nfs_advect2d_bwd", _ptr(d), _ptr(vel), _ptr(g_out), _ptr(g_d), _ptr(g_vel), H, W, Cn, _stream())
    return g_d, g_vel


def advect_maccormack(d, vel):
    """order-2 (MacCormack) advection with the extrema limiter (transform.py:570-582, 590-607 as intended);
    d [D,H,W,C] + vel [D,H,W,3], or d [H,W,C] + vel [H,W,2].  Forward only."""
    nd = vel.shape[-1]
    if nd == 3:
        D, H, W, Cn = d.shape
        d_fwd = advect_fwd(d, vel)
    else:
        (H, W, Cn), D = d.shape, 1
        d_fwd = advect2d_fwd(d, vel)
    out = _empty(d.shape, d)
    _lib.call("nfs_advect_maccormack", _ptr(d), _ptr(vel), _ptr(d_fwd), _ptr(out), D, H, W, Cn, nd, _stream())
    return out


def curl_fwd(s):
    """s [H,W] -> [H,W,2] or s [D,H,W,3] -> [D,H,W,3]  (transform.py:517-555)"""
    if s.dim() == 2:
        (H, W), D, nd = s.shape, 1, 2
        out = _empty((H, W, 2), s)
    else:
        (D, H, W, _), nd = s.shape, 3
        out = _empty(s.shape, s)
    _lib.call("nfs_curl_fwd", _ptr(s), _ptr(out), D, H, W, nd, _stream())
    return out


def curl_bwd(g_out):
    if g_out.shape[-1] == 2:
        (H, W, _), D, nd = g_out.shape, 1, 2
        g_s = _empty((H, W), g_out)
    else:
        (D, H, W, _), nd = g_out.shape, 3
        g_s = _empty(g_out.shape, g_out)
    _lib.call("nfs_curl_bwd", _ptr(g_out), _ptr(g_s), D, H, W, nd, _stream())
    return g_s


def lap_down(x, k):
    """x [D,H,W,C] (3-D, k [5,5,5]) or [H,W,C] (2-D, k [5,5]) -> stride-2 'SAME' smoothing (util.py:60-66)"""
    nd = 3 if x.dim() == 4 else 2
    D, (H, W, Cn) = (x.shape[0] if nd == 3 else 1), x.shape[-3:]
    shp = (((D + 1) // 2,) if nd == 3 else ()) + ((H + 1) // 2, (W + 1) // 2, Cn)
    out = _empty(shp, x)
    _lib.call("nfs_lap_down", _ptr(x), _ptr(k), _ptr(out), D, H, W, Cn, nd, _stream())
    return out


def lap_up(lo, k, out_shape, scale, addend=None):
    """scale * conv_transpose(lo, k, out_shape, stride 2) + addend (util.py:62-65, 80-83)"""
    nd = 3 if lo.dim() == 4 else 2
    out = _empty(tuple(out_shape), lo)
    D, (H, W, Cn) = (out_shape[0] if nd == 3 else 1), tuple(out_shape)[-3:]
    _lib.call("nfs_lap_up", _ptr(lo), _ptr(k), float(scale), _ptr(addend), _ptr(out), D, H, W, Cn, nd, _stream())
    return out


def lap_up_rms_parts(out_shape):
    """partial sums ``lap_up_rms`` writes for an output of this shape; 0 where only ``lap_up`` applies"""
    nd = 3 if len(out_shape) == 4 else 2
    D, (H, W, Cn) = (out_shape[0] if nd == 3 else 1), tuple(out_shape)[-3:]
    return int(_lib.lib().nfs_lap_up_rms_parts(int(D), int(H), int(W), int(Cn), nd))


def lap_up_rms(lo, k, out_shape, scale, addend=None, addend_part=None, eps=1e-10, want_part=False):
    """``lap_up`` with the RMS normalisation of pyramid levels riding in it (nfs_hip.h: nfs_lap_up_rms): ``addend_part``
    (the partial sums of addend^2 an earlier call returned): the addend enters divided by max(rms(addend), eps);
    ``want_part``: also return the partial sums of out^2.  -> out, or (out, part)"""
    nd = 3 if lo.dim() == 4 else 2
    out = _empty(tuple(out_shape), lo)
    D, (H, W, Cn) = (out_shape[0] if nd == 3 else 1), tuple(out_shape)[-3:]
    part = _empty((lap_up_rms_parts(out_shape),), lo) if want_part else None
    _lib.call("nfs_lap_up_rms", _ptr(lo), _ptr(k), float(scale), _ptr(addend), _ptr(addend_part),
              0 if addend_part is None else addend_part.numel(), 0 if addend is None else addend.numel(), float(eps),
              _ptr(out), _ptr(part), D, H, W, Cn, nd, _stream())
    return (out, part) if want_part else out


def normalize_mean(x, use_abs=False, eps=1e-10):
    """x / max(sqrt(mean(x^2)), eps)  (normalize_std, util.py:86-90)  or  x / max(mean|x|, eps)"""
    out = _empty(x.shape, x)
    ws = _empty((1024,), x)
    _lib.call("nfs_normalize_mean", _ptr(x), _ptr(out), x.numel(), int(bool(use_abs)), float(eps), _ptr(ws), 1024,
              _stream())
    return out


def transport_step(g, u, scale=1.0, w_g=1.0, addend=None, w_addend=0.0, out=None):
    """out = w_g*advect(g, scale*u) + w_addend*addend: one frame crossing of ``_transport`` (styler_base.py:59-89) for the
    C-channel field g [D,H,W,C] with the temporal filter's accumulation fused in"""
    D, H, W, Cn = g.shape
    if out is None:
        out = _empty(g.shape, g)
    _lib.call("nfs_transport_step", _ptr(g), _ptr(u), float(scale), float(w_g), _ptr(addend), float(w_addend), _ptr(out),
              D, H, W,
              Cn, _stream())
    return out


# ---- A9 -----------------------------------------------------------------------------

def smooth3d_relu_fwd(d, k, out=None):
    D, H, W = d.shape
    if out is None:
        out = _empty(d.shape, d)
    _lib.call("nfs_smooth3d_relu_fwd", _ptr(d), _ptr(out), D, H, W, float(k), _stream())
    return out


def smooth3d_relu_bwd(out, g_out, k, g_d=None):
    D, H, W = out.shape
    if g_d is None:
        g_d = _empty(out.shape, out)
    _lib.call("nfs_smooth3d_relu_bwd", _ptr(out), _ptr(g_out), _ptr(g_d), D, H, W, float(k), _stream())
    return g_d


# ---- A4 -----------------------------------------------------------------------------

def render_fwd(d, tau, liquid=False):
    """d [V,D,H,W] -> (img [V,H,W] un-normalised, raysum [V,H,W])"""
    V, D, H, W = d.shape
    img = _empty((V, H, W), d); rs = _empty((V, H, W), d)
    _lib.call("nfs_render_fwd", _ptr(d), _ptr(img), _ptr(rs), V, D, H, W, float(tau), int(liquid), _stream())
    return img, rs


def render_bwd(d, raysum, g_img, tau, liquid=False, g_d=None, want_max=False):
    """want_max: also return the device scalar max|g_d| (a by-product of the same pass) for rotate_bwd(g_max=...)"""
    V, D, H, W = d.shape
    if g_d is None:
        g_d = _empty(d.shape, d)
    gmax = _empty((1,), d) if want_max else None
    _lib.call("nfs_render_bwd", _ptr(d), _ptr(raysum), _ptr(g_img), _ptr(g_d), V, D, H, W, float(tau),
              int(liquid), _ptr(gmax), _stream())
    return (g_d, gmax) if want_max else g_d


def rotate_render_fwd(d, rot, tau, liquid=False, img=None, raysum=None, d_rot=None):
    """d [D,H,W], rot [V,3,3] -> (img [V,H,W], raysum [V,H,W]); d_rot [V,D,H,W] (optional) is filled
    with the rotated samples for the two-pass adjoint"""
    D, H, W = d.shape
    V = rot.shape[0]
    if img is None:
        img = _empty((V, H, W), d)
    if raysum is None:
        raysum = _empty((V, H, W), d)
    _lib.call("nfs_rotate_render_fwd", _ptr(d), _ptr(rot), _ptr(img), _ptr(raysum), _ptr(d_rot), V, D, H, W,
              float(tau), int(liquid), _stream())
    return img, raysum


_coef_layouts = {}


def render_coef_layout(V, D, H, W):
    """(nseg, seg_len) of the u / coefficient form of the rotate + render adjoint (transmittance mode), or None when the
    shape is outside the segmented forward kernel -- the caller then keeps the rotated samples and runs render_bwd"""
    key = (int(V), int(D), int(H), int(W))
    if key not in _coef_layouts:
        import ctypes
        nseg, seg_len = ctypes.c_int(0), ctypes.c_int(0)
        rc = _lib.lib().nfs_render_coef_layout(key[0], key[1], key[2], key[3], ctypes.byref(nseg), ctypes.byref(seg_len))
        _coef_layouts[key] = (nseg.value, seg_len.value) if rc == 0 else None
    return _coef_layouts[key]


def rotate_render_fwd_coef(d, rot, tau, img=None, raysum=None, u_rot=None, seg=None):
    """d [D,H,W], rot [V,3,3] -> (img [V,H,W], raysum [V,H,W], u_rot [V,D,H,W], seg [3,V,nseg,H,W]): the forward of
    rotate_render_fwd (transmittance) that keeps u = t + tau i per sample instead of the sample (see nfs_hip.h)"""
    D, H, W = d.shape
    V = rot.shape[0]
    nseg, _ = render_coef_layout(V, D, H, W)
    if img is None:
        img = _empty((V, H, W), d)
    if raysum is None:
        raysum = _empty((V, H, W), d)
    if u_rot is None:
        u_rot = _empty((V, D, H, W), d)
    if seg is None:
        seg = _empty((3, V, nseg, H, W), d)
    _lib.call("nfs_rotate_render_fwd_coef", _ptr(d), _ptr(rot), _ptr(img), _ptr(raysum), _ptr(u_rot), _ptr(seg), V, D, H, W,
              float(tau), _stream())
    return img, raysum, u_rot, seg


def render_ray_coef(g_img, seg, tau, ab=None, bounds=None):
    """image gradient g_img [V,H,W] + the forward's seg -> (ab [V,nseg,H,W,2], bounds [nblocks]: per-block bounds on
    max |sample gradient|, reduced by the consumer)"""
    V, H, W = g_img.shape
    nseg = seg.shape[2]
    if ab is None:
        ab = _empty((V, nseg, H, W, 2), g_img)
    if bounds is None:
        bounds = _empty((_lib.lib().nfs_render_ray_coef_bounds(V, H, W),), g_img)
    _lib.call("nfs_render_ray_coef", _ptr(g_img), _ptr(seg), _ptr(ab), _ptr(bounds), V, H, W, float(tau), _stream())
    return ab, bounds


_LIVE_WS = {}      # (device, D, H, W, stream) -> workspace of nfs_rotate_bwd_coef_live


def rotate_bwd_coef(u_rot, ab, rot, bounds, g_d_acc=None, overwrite=False, live=None, dilate=1):
    """the tiled rotate adjoint with the sample gradient A u - B formed on the fly; g_d_acc [D,H,W] (+=, or written).
    ``live`` (``advect_fwd(live=...)``): only the voxels within ``dilate`` cells of a live voxel get their sums (the rest
    of g_d is multiplied by an exact zero in the advect adjoint): tiles without any return at once"""
    V, D, H, W = u_rot.shape
    nseg, seg_len = render_coef_layout(V, D, H, W)
    if g_d_acc is None:
        overwrite = True
        g_d_acc = _empty((D, H, W), u_rot)
    if live is not None:
        # (one per stream: view groups on side streams run this adjoint concurrently, and a launch owns its ticket)
        key = (u_rot.device, D, H, W, int(_stream(u_rot.device) or 0))
        ws = _LIVE_WS.get(key)
        if ws is None:      # zeroed once: every launch leaves its ticket zero again (and a captured graph keeps the address)
            ws = _LIVE_WS[key] = _zeros((int(_lib.lib().nfs_rotate_live_workspace_ints(D, H, W)),), u_rot)
        _lib.call("nfs_rotate_bwd_coef_live", _ptr(u_rot), _ptr(ab), _ptr(rot), _ptr(g_d_acc), V, D, H, W, nseg, seg_len,
                  _ptr(bounds), int(bounds.numel()), int(bool(overwrite)), _ptr(live), int(dilate), _ptr(ws), _stream())
    else:
        _lib.call("nfs_rotate_bwd_coef", _ptr(u_rot), _ptr(ab), _ptr(rot), _ptr(g_d_acc), V, D, H, W, nseg, seg_len,
                  _ptr(bounds), int(bounds.numel()), int(bool(overwrite)), _stream())
    return g_d_acc


def rotate_render_bwd(d, rot, raysum, g_img, tau, liquid=False, g_d_acc=None):
    D, H, W = d.shape
    V = rot.shape[0]
    if g_d_acc is None:
        g_d_acc = _zeros(d.shape, d)
    _lib.call("nfs_rotate_render_bwd", _ptr(d), _ptr(rot), _ptr(raysum), _ptr(g_img), _ptr(g_d_acc), V, D, H, W,
              float(tau), int(liquid), _stream())
    return g_d_acc


def maxnorm_fwd(img, groups, out=None, gmax=None):
    n = img.numel() // groups
    if out is None:
        out = _empty(img.shape, img)
    if gmax is None:
        gmax = _empty((groups,), img)
    _lib.call("nfs_maxnorm_fwd", _ptr(img), _ptr(out), _ptr(gmax), groups, n, _stream())
    return out, gmax


def maxnorm_bwd(img, gmax, g_out, g_img=None):
    groups = gmax.numel()
    n = img.numel() // groups
    if g_img is None:
        g_img = _empty(img.shape, img)
    ws = _empty((64 * groups,), img)            # partial sums of the multi-block path
    _lib.call("nfs_maxnorm_bwd", _ptr(img), _ptr(gmax), _ptr(g_out), _ptr(g_img), groups, n, _ptr(ws), _stream())
    return g_img


def maxnorm_input_fwd(img, groups):
    """img [V,H,W] (grey render) -> (x [V,H,W,3] = (img / max) * 255 - mean, gmax [groups]): nfs_maxnorm_fwd +
    nfs_loss_net_input_fwd in one pass (loss net at the render's size)"""
    V, H, W = img.shape
    x = _empty((V, H, W, 3), img)
    gmax = _empty((groups,), img)
    _lib.call("nfs_maxnorm_input_fwd", _ptr(img), _ptr(x), _ptr(gmax), groups, img.numel() // groups, _stream())
    return x, gmax


def maxnorm_input_bwd(img, gmax, g_x):
    """adjoint of maxnorm_input_fwd: g_x [V,H,W,3] -> g_img [V,H,W]"""
    groups = gmax.numel()
    g_img = _empty(img.shape, img)
    ws = _empty((64 * groups,), img)
    _lib.call("nfs_maxnorm_input_bwd", _ptr(img), _ptr(gmax), _ptr(g_x), _ptr(g_img), groups, img.numel() // groups,
              _ptr(ws), _stream())
    return g_img


# ---- A5 -----------------------------------------------------------------------------

def loss_net_input_fwd(img, H2=None, W2=None, want_d_img=True, want_x=True):
    """img [B,H,W,Cin] in [0,1] -> (d_img [B,H2,W2,3] 0..255, x = d_img - mean)"""
    B, H, W, Cin = img.shape
    H2 = H if H2 is None else H2
    W2 = W if W2 is None else W2
    d_img = _empty((B, H2, W2, 3), img) if want_d_img else None
    x = _empty((B, H2, W2, 3), img) if want_x else None
    _lib.call("nfs_loss_net_input_fwd", _ptr(img), _ptr(d_img), _ptr(x), B, H, W, Cin, H2, W2, _stream())
    return d_img, x


def loss_net_input_bwd(g_x, H, W, Cin):
    B, H2, W2, _ = g_x.shape
    g_img = _empty((B, H, W, Cin), g_x)
    _lib.call("nfs_loss_net_input_bwd", _ptr(g_x), _ptr(g_img), B, H, W, Cin, H2, W2, _stream())
    return g_img


# ---- A6 -----------------------------------------------------------------------------

def gemm_mode(mode=None):
    """Arithmetic of the batched Winograd GEMMs (process-wide): 1 = split-limb form, the default (float32 operands as
    three exact bf16 limbs, six limb products on the bf16 MFMA, float32 accumulation: float32-equivalent accuracy at 2.67x
    the matrix-pipe rate), 0 = float32-input MFMA.  Returns the previous mode; ``None`` only queries."""
    return int(_lib.lib().nfs_gemm_mode(-1 if mode is None else int(mode)))


def conv3x3_pack(w_hwio, kind):
    """w [3,3,Ci,Co] -> packed device buffer for kind 0 (fwd) / 1 (dgrad)"""
    _, _, Ci, Co = w_hwio.shape
    n = _lib.lib().nfs_conv3x3_packed_floats(Ci, Co, kind)
    packed = _empty((n,), w_hwio)
    _lib.call("nfs_conv3x3_pack", _ptr(w_hwio), _ptr(packed), Ci, Co, kind, _stream())
    return packed


_conv_ws = {}


def conv_workspace(device, floats):
    """split-K scratch, grown on demand and shared by all conv calls of a device (calls on one stream
    are ordered, so sharing is safe)"""
    if torch.cuda.is_current_stream_capturing():
        # (see ``workspace``: a capture owns its scratch.)  A fresh buffer per call: the graph's private pool hands the
        # block back to later allocations of the same capture in stream order and keeps it for the replays
        return torch.empty(int(floats), dtype=torch.float32, device=device)
    key = (device.type, device.index, _stream(device))   # one per stream
    cur = _conv_ws.get(key)
    if cur is None or cur.numel() < floats:
        cur = torch.empty(int(floats), dtype=torch.float32, device=device)
        _conv_ws[key] = cur
    return cur


def _conv_ws_for(B, H, W, Ci, Co, device, splitk):
    if not splitk:
        return None, 0
    want = _lib.lib().nfs_conv3x3_workspace_floats(B, H, W, Ci, Co)   # split-K partials or Winograd V/M buffers
    ws = conv_workspace(device, want)
    return ws, ws.numel()


def conv3x3_relu_bits_words(B, H, W, Ci, Co, pooled):
    return int(_lib.lib().nfs_conv3x3_relu_bits_words(B, H, W, Ci, Co, int(bool(pooled))))


def conv3x3_relu_bits(B, H, W, Ci, Co, pooled, device):
    """buffer for a layer's ReLU bit cache (raw 32-bit words, held in a float32 tensor), or None when the layer does
    not keep one: pass it to the layer's conv3x3_fwd / conv3x3_fwd_pool and later to its conv3x3_dgrad / _pool"""
    n = _lib.lib().nfs_conv3x3_relu_bits_words(B, H, W, Ci, Co, int(bool(pooled)))
    return torch.empty(n, dtype=torch.float32, device=device) if n > 0 else None


def conv3x3_fwd(x, packed, bias, Co, relu=True, out=None, splitk=True, relu_bits=None):
    B, H, W, Ci = x.shape
    if out is None:
        out = _empty((B, H, W, Co), x)
    ws, nws = _conv_ws_for(B, H, W, Ci, Co, x.device, splitk)
    _lib.call("nfs_conv3x3_fwd", _ptr(x), _ptr(packed), _ptr(bias), _ptr(out), B, H, W, Ci, Co, int(relu),
              _ptr(ws), nws, _ptr(relu_bits if splitk else None), _stream())
    return out


def conv3x3_dgrad(gy, packed, Ci, x_in=None, addend=None, out=None, splitk=True, relu_bits=None,
                  addend_unmasked=False):
    B, H, W, Co = gy.shape
    if out is None:
        out = _empty((B, H, W, Ci), gy)
    ws, nws = _conv_ws_for(B, H, W, Ci, Co, gy.device, splitk)
    _lib.call("nfs_conv3x3_dgrad", _ptr(gy), _ptr(packed), _ptr(x_in), _ptr(addend), _ptr(out), B, H, W, Ci, Co,
              _ptr(ws), nws, _ptr(relu_bits if (splitk and x_in is not None) else None), int(bool(addend_unmasked)),
              _stream())
    return out


def conv3x3_fwd_pool(x, packed, bias, Co, relu=True, relu_bits=None, want_y=True):
    """conv + bias + ReLU and its 2x2 VALID average pool in one pass -> (y [B,H,W,Co], y_pool [B,H/2,W/2,Co]);
    with a ReLU bit cache and want_y=False the full-resolution output is not written at all (y = None)"""
    B, H, W, Ci = x.shape
    out = _empty((B, H, W, Co), x) if (want_y or relu_bits is None) else None
    pooled = _empty((B, H // 2, W // 2, Co), x)
    ws, nws = _conv_ws_for(B, H, W, Ci, Co, x.device, True)
    _lib.call("nfs_conv3x3_fwd_pool", _ptr(x), _ptr(packed), _ptr(bias), _ptr(out), _ptr(pooled), B, H, W, Ci, Co,
              int(relu), _ptr(ws), nws, _ptr(relu_bits), _stream())
    return out, pooled


def conv3x3_dgrad_pool(gy_pool, x_out, packed, Ci, x_in=None, addend=None, relu_bits=None, hw=None,
                       addend_unmasked=False):
    """data gradient of a conv followed by ReLU + 2x2 average pool, from the gradient at the POOLED resolution
    gy_pool [B,H/2,W/2,Co] and the conv's own output x_out [B,H,W,Co] -> gx [B,H,W,Ci]; x_out may be None when the
    layer's ReLU bit cache is given (``hw`` = its (H, W) then)"""
    if x_out is not None:
        B, H, W, Co = x_out.shape
    else:
        assert relu_bits is not None and hw is not None
        B, Co = gy_pool.shape[0], gy_pool.shape[-1]
        H, W = hw
    out = _empty((B, H, W, Ci), gy_pool)
    ws, nws = _conv_ws_for(B, H, W, Ci, Co, gy_pool.device, True)
    _lib.call("nfs_conv3x3_dgrad_pool", _ptr(gy_pool), _ptr(x_out), _ptr(packed), _ptr(x_in), _ptr(addend), _ptr(out),
              B, H, W, Ci, Co, _ptr(ws), nws, _ptr(relu_bits), int(bool(addend_unmasked)), _stream())
    return out


def avgpool2_fwd(x, out=None):
    B, H, W, Cn = x.shape
    if out is None:
        out = _empty((B, H // 2, W // 2, Cn), x)
    _lib.call("nfs_avgpool2_fwd", _ptr(x), _ptr(out), B, H, W, Cn, _stream())
    return out


def avgpool2_bwd(gy, x_shape, x=None, addend=None, out=None):
    B, H, W, Cn = x_shape
    if out is None:
        out = _empty(tuple(x_shape), gy)
    _lib.call("nfs_avgpool2_bwd", _ptr(gy), _ptr(x), _ptr(addend), _ptr(out), B, H, W, Cn, _stream())
    return out


# ---- A7 / A12 -----------------------------------------------------------------------

def gram_fwd(F, scale, scale_dev=None, G=None, two_pass=True):
    """F [B,h,w,C] (or [B,HW,C]) -> G [B,C,C] = scale * F^T F"""
    B, Cn = F.shape[0], F.shape[-1]
    HW = F.numel() // (B * Cn)
    ws, nws = None, 0
    if two_pass:
        nws = _lib.lib().nfs_gram_workspace_floats(B, HW, Cn)
        ws = conv_workspace(F.device, nws)      # shared scratch (calls on one stream are ordered)
        nws = ws.numel()
    if G is None:
        G = _empty((B, Cn, Cn), F) if two_pass else _zeros((B, Cn, Cn), F)
    elif not two_pass:
        G.zero_()
    _lib.call("nfs_gram_fwd", _ptr(F), _ptr(G), B, HW, Cn, _ptr(scale_dev), float(scale), _ptr(ws), nws, _stream())
    return G


def style_loss_fwd(G, Gs, weight, loss_acc, Dmat=None):
    B, Cn, _ = G.shape
    if Dmat is None:
        Dmat = _empty(G.shape, G)
    _lib.call("nfs_style_loss_fwd", _ptr(G), _ptr(Gs), _ptr(loss_acc), _ptr(Dmat), B, Gs.shape[0], Cn, float(weight),
              _stream())
    return Dmat


def gram_style_group(Fs, Gss, weights, relu_masks, want_G=False, channels=None):
    """Gram matrix, style loss and Gram gradient of SEVERAL style layers in three launches (the loop of
    styler_base.py:152-185): ``Fs`` list of [B,h,w,C] activations, ``Gss`` their style Grams [Bs,C,C] (scaled by
    1/(2 h w C) like G), ``weights`` w_layer * w_style, ``relu_masks`` whether dF carries the layer's ReLU mask.
    Returns (loss_parts [P,B] -- the style loss of image b is loss_parts[:, b].sum(), every entry written, no atomics --,
    list of dF, list of G or None)."""
    import ctypes
    n = len(Fs)
    if n > 8:                                          # the descriptor table of one launch holds 8 layers
        a = gram_style_group(Fs[:8], Gss[:8], weights[:8], relu_masks[:8], want_G, channels and channels[:8])
        b = gram_style_group(Fs[8:], Gss[8:], weights[8:], relu_masks[8:], want_G, channels and channels[8:])
        return torch.cat([a[0], b[0]]), a[1] + b[1], (a[2] + b[2] if want_G else None)
    arr = (_lib.GramLayer * n)()
    keep = []
    B = Fs[0].shape[0]
    for l, (F, Gs, w, rm) in enumerate(zip(Fs, Gss, weights, relu_masks)):
        Cn = F.shape[-1]
        HW = F.numel() // (B * Cn)
        Dm = _empty((B, Cn, Cn), F)
        dF = _empty(F.shape, F)
        G = _empty((B, Cn, Cn), F) if want_G else None
        keep.append((Dm, dF, G))
        y = arr[l]
        y.F, y.Gs, y.G, y.Dmat, y.dF = _ptr(F), _ptr(Gs), _ptr(G), _ptr(Dm), _ptr(dF)
        y.B, y.Bs, y.HW, y.C = B, Gs.shape[0], HW, Cn
        # (rows padded with zero channels -- Inception module outputs --: the denominator counts the logical ones)
        y.scale, y.weight, y.relu_mask = 1.0 / (2.0 * HW * (channels[l] if channels else Cn)), float(w), int(bool(rm))
    L = _lib.lib()
    ap = ctypes.cast(arr, ctypes.c_void_p)
    P = L.nfs_gram_style_group_parts(ap, n)
    nws = L.nfs_gram_style_group_workspace_floats(ap, n)
    if P < 0 or nws < 0:
        raise ValueError("gram_style_group: 1..8 layers of one batch size with C a multiple of 64")
    ws = conv_workspace(Fs[0].device, max(nws, 1))      # shared scratch (calls on one stream are ordered)
    parts = _empty((P, B), Fs[0])
    _lib.call("nfs_gram_style_group_fwd", ap, n, _ptr(parts), _ptr(ws), ws.numel(), _stream())
    _lib.call("nfs_gram_group_bwd", ap, n, _stream())
    return parts, [k[1] for k in keep], ([k[2] for k in keep] if want_G else None)


def content_loss(F, weight, loss_acc, g_acc, channel=0, target=None, amp=100.0, signed=False):
    """content loss of the post-ReLU activation F [B,h,w,C] (styler_base.py:135-150): with ``target`` [Bt,h,w,C]
    mean((F - amp*target)^2), else channel maximisation (channel != 0) or -mean(F); loss_acc [B] and
    g_acc [B,h,w,C] (gradient wrt the pre-activation) are accumulated into"""
    B, Cn = F.shape[0], F.shape[-1]
    HW = F.numel() // (B * Cn)
    mode = 2 if target is not None else (0 if channel else 1)
    Bt = target.shape[0] if target is not None else 0
    # ``signed``: F is not a ReLU output (an Inception '*_pre_relu' tensor): |F| proper, gradient wrt F itself
    _lib.call("nfs_content_loss_signed" if signed else "nfs_content_loss", _ptr(F), _ptr(target), _ptr(loss_acc),
              _ptr(g_acc), B, Bt, HW, Cn, int(channel or 0),
              mode, float(weight), float(amp), _stream())
    return g_acc


def resize_bicubic_tf1(x, oh, ow):
    """TF-1 legacy bicubic resize of x [B,H,W,C] -> [B,oh,ow,C] (styler_base.py:166), forward only"""
    B, H, W, Cn = x.shape
    out = _empty((B, oh, ow, Cn), x)
    _lib.call("nfs_resize_bicubic_tf1", _ptr(x), _ptr(out), B, H, W, Cn, int(oh), int(ow), _stream())
    return out


def style_mask_apply(F, mask):
    """F [B,h,w,C], mask [B,h,w,1] -> (F * mask, scale [B] = 1 / (2 * sum(mask) * C))  (styler_base.py:167-169)"""
    B, Cn = F.shape[0], F.shape[-1]
    HW = F.numel() // (B * Cn)
    Fm = _empty(F.shape, F)
    scale = _empty((B,), F)
    _lib.call("nfs_style_mask_apply", _ptr(F), _ptr(mask), _ptr(Fm), _ptr(scale), B, HW, Cn, _stream())
    return Fm, scale


def style_mask_bwd(dFm, mask, F):
    """dFm * mask * (F > 0)"""
    B, Cn = F.shape[0], F.shape[-1]
    HW = F.numel() // (B * Cn)
    out = _empty(F.shape, F)
    _lib.call("nfs_style_mask_bwd", _ptr(dFm), _ptr(mask), _ptr(F), _ptr(out), B, HW, Cn, _stream())
    return out


import os as _os
_HIST_WIDE = _os.environ.get("NFS_HIST_WIDE", "1") != "0"   # 0: per-channel kernel for every C > 4 (timing comparisons)


def hist_loss(F, templ, weight, loss_acc, g_acc=None, relu_mask=False, mask=None):
    """histogram loss of F [B,h,w,C] against the template features templ [Bt,ht,wt,C] (styler_base.py:187-209,
    util.py:317-399): loss_acc [B] += weight * sum((F - matched)^2); g_acc [B,h,w,C] += 2 weight (F - matched).
    ``mask`` [B,h,w(,1)]: the masked branch (styler_base.py:104-125, 196-201) -- pixels where it is 0 leave the source."""
    B, Cn = F.shape[0], F.shape[-1]
    HW = F.numel() // (B * Cn)
    Bt = templ.shape[0]
    HWt = templ.numel() // (Bt * Cn)
    assert templ.shape[-1] == Cn
    if mask is not None:
        assert mask.numel() == B * HW and mask.is_contiguous(), "mask must be [B,h,w] of the feature's size"
    if Cn <= 4 or (Cn % 4 == 0 and Cn <= 128 and _HIST_WIDE):
        # few channels (the default hist layer is the 3-channel loss-net input): one block per (image, channel) would
        # leave the chip idle -- the pixel-parallel form, its per-channel state in a workspace (300 x 450 x 3: 0.73 ->
        # 0.075 ms, 512 x 1024 x 3: 3.9 -> 0.10 ms).  Up to 128 channels it also beats the per-channel kernel's
        # channel-strided reads (150 x 225 x 64: 0.37 -> 0.21 ms); beyond that there are enough (image, channel) blocks
        nws = _lib.lib().nfs_hist_loss_wide_workspace_floats(B, Cn, HW, HWt)
        ws = conv_workspace(F.device, nws)
        _lib.call("nfs_hist_loss_wide", _ptr(F), _ptr(templ), _ptr(mask), _ptr(loss_acc), _ptr(g_acc), _ptr(ws), nws, B,
                  Bt, HW, HWt, Cn, float(weight), int(bool(relu_mask)), _stream())
        return g_acc
    if mask is not None:
        _lib.call("nfs_hist_loss_masked", _ptr(F), _ptr(templ), _ptr(mask), _ptr(loss_acc), _ptr(g_acc), B, Bt, HW, HWt,
                  Cn, float(weight), int(bool(relu_mask)), _stream())
        return g_acc
    _lib.call("nfs_hist_loss", _ptr(F), _ptr(templ), _ptr(loss_acc), _ptr(g_acc), B, Bt, HW, HWt, Cn, float(weight),
              int(bool(relu_mask)), _stream())
    return g_acc


def gram_bwd(F, Dmat, scale, scale_dev=None, relu_mask=True, out=None):
    B, Cn = F.shape[0], F.shape[-1]
    HW = F.numel() // (B * Cn)
    if out is None:
        out = _empty(F.shape, F)
    _lib.call("nfs_gram_bwd", _ptr(F), _ptr(Dmat), _ptr(out), B, HW, Cn, _ptr(scale_dev), float(scale),
              int(relu_mask), _stream())
    return out


def tv_loss(d_img, weight, loss_acc, g_acc=None):
    B, H, W, Cn = d_img.shape
    _lib.call("nfs_tv_loss", _ptr(d_img), _ptr(loss_acc), _ptr(g_acc), B, H, W, Cn, float(weight), _stream())


# ---- A8 -----------------------------------------------------------------------------

def make_splat_cfg(nd, res, domain, radius, support, rest_density, nsize, clip, mode):
    c = SplatCfg()
    c.nd = nd
    for k in range(3):
        c.res[k] = int(res[k]) if k < nd else 1
        c.domain[k] = float(domain[k]) if k < nd else 1.0
    c.radius = float(radius); c.support = float(support); c.rest_density = float(rest_density)
    c.nsize = int(nsize); c.clip = int(bool(clip)); c.mode = int(mode)
    return c


def p2g_fwd(p, cfg, attr=None, pd=None):
    """p [N,nd]; returns grid [res..., C] (mode 0/1) or (xsum, wsum) (mode 2)"""
    N = p.shape[0]
    Cn = 1 if attr is None else attr.shape[-1]
    res = [cfg.res[k] for k in range(cfg.nd)]
    grid = _zeros(tuple(res) + (Cn,), p)
    wsum = _zeros(tuple(res) + (1,), p) if cfg.mode == 2 else None
    _lib.call("nfs_p2g_fwd", _ptr(p), _ptr(attr), _ptr(pd), _ptr(grid), _ptr(wsum), N, Cn, C.byref(cfg), _stream())
    return (grid, wsum) if cfg.mode == 2 else grid


def p2g_bwd(p, cfg, g_grid, attr=None, pd=None, g_wsum=None, need_p=True, need_attr=False, need_pd=False):
    N = p.shape[0]
    Cn = 1 if attr is None else attr.shape[-1]
    g_p = _empty(p.shape, p) if need_p else None
    g_attr = _empty(attr.shape, p) if need_attr else None
    g_pd = _empty((N,), p) if need_pd else None
    _lib.call("nfs_p2g_bwd", _ptr(p), _ptr(attr), _ptr(pd), _ptr(g_grid), _ptr(g_wsum), _ptr(g_p), _ptr(g_attr),
              _ptr(g_pd), N, Cn, C.byref(cfg), _stream())
    return g_p, g_attr, g_pd


def p2g_wavg_bwd(p, cfg, xsum, wsum, g_out, attr, eps=1e-6, need_p=True, need_attr=True):
    """adjoint of the weighted-average splat in one launch (finish adjoint folded into the gather); None when the
    neighbourhood has no compile-time instance (the caller takes p2g_wavg_finish_bwd + p2g_bwd)"""
    if not ((cfg.nd == 3 and cfg.nsize in (1, 2)) or (cfg.nd == 2 and cfg.nsize in (1, 2, 3, 4))):
        return None
    N = p.shape[0]
    Cn = attr.shape[-1]
    g_p = _empty(p.shape, p) if need_p else None
    g_attr = _empty(attr.shape, p) if need_attr else None
    try:
        _lib.call("nfs_p2g_wavg_bwd", _ptr(p), _ptr(attr), _ptr(xsum), _ptr(wsum), _ptr(g_out), _ptr(g_p), _ptr(g_attr), N,
                  Cn, float(eps), C.byref(cfg), _stream())
    except _lib.NfsError as e:
        # the library's own predicate (launch_p2g_bwd: cells < 2^31, its reading of NFS_SPLAT_LDS) is the authority: the
        # ONE refusal it documents -- no compile-time instance, nothing launched -- sends the caller down the two-launch
        # adjoint; any other NFS_EINVAL (a null pointer, a bad N or mode) is an error, not a slow path
        if e.code != _lib.NFS_EINVAL or "no compile-time instance" not in str(e):
            raise
        return None
    return g_p, g_attr


def p2g_wavg_finish(xsum, wsum, eps=1e-6):
    Cn = xsum.shape[-1]
    n = wsum.numel()
    out = _empty(xsum.shape, xsum)
    _lib.call("nfs_p2g_wavg_finish", _ptr(xsum), _ptr(wsum), _ptr(out), n, Cn, float(eps), _stream())
    return out


def p2g_wavg_finish_bwd(xsum, wsum, g_out, eps=1e-6):
    Cn = xsum.shape[-1]
    n = wsum.numel()
    g_x = _empty(xsum.shape, xsum); g_w = _empty(wsum.shape, xsum)
    _lib.call("nfs_p2g_wavg_finish_bwd", _ptr(xsum), _ptr(wsum), _ptr(g_out), _ptr(g_x), _ptr(g_w), n, Cn,
              float(eps), _stream())
    return g_x, g_w


# ---- A10 ----------------------------------------------------------------------------

def _written(*tensors):
    """the kernels write through raw pointers: tell torch's version counters (views share them), so that anything keyed on
    ``tensor._version`` -- the stylizer's stored forward advect, autograd's saved-tensor checks -- sees the write"""
    for t in tensors:
        if t is not None:
            torch.autograd.graph.increment_version(t)


def adam_tf_step(x, m, v, g, lr_t, beta1=0.9, beta2=0.999, eps=1e-8):
    _lib.call("nfs_adam_tf_step", _ptr(x), _ptr(m), _ptr(v), _ptr(g), x.numel(), float(lr_t), float(beta1),
              float(beta2), float(eps), _stream())
    _written(x, m, v)


def slab_pack(gpad, pack, D, world, cs):
    """send buffer of the D-slab reduce-scatter: pack [world, cs+5, H, W] <- overlapping plane ranges of gpad [D+5, H, W]
    (nfs_hip.h: nfs_slab_pack)"""
    assert gpad.is_contiguous() and pack.is_contiguous() and gpad.shape[0] == D + 5
    assert tuple(pack.shape) == (world, cs + 5) + tuple(gpad.shape[1:])
    _lib.call("nfs_slab_pack", _ptr(gpad), _ptr(pack), int(D), int(gpad[0].numel()), int(world), int(cs), _stream())
    return pack


def _iptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.int64 and t.is_contiguous()
    return t.data_ptr()


def colour_clamp_gather(var, order=None):
    """clip(var, 0, 1) read through a permutation (int64 [N], None = identity): var [N,C] -> [N,C]"""
    out = _empty(var.shape, var)
    _lib.call("nfs_colour_clamp_gather", _ptr(var), _iptr(order), _ptr(out), var.shape[0], var.shape[1], _stream())
    return out


def clamp01_bwd(g, x):
    """adjoint of clip(x, 0, 1): g where 0 <= x <= 1, else 0"""
    out = _empty(g.shape, g)
    _lib.call("nfs_clamp01_bwd", _ptr(g), _ptr(x), _ptr(out), g.numel(), _stream())
    return out


def colour_clamp_scatter_bwd(g_cc, var, order=None):
    """adjoint of colour_clamp_gather: g_var[order[i]] = g_cc[i] where 0 <= var[order[i]] <= 1, else 0"""
    g_var = _empty(var.shape, var)
    _lib.call("nfs_colour_clamp_scatter_bwd", _ptr(g_cc), _iptr(order), _ptr(var), _ptr(g_var), var.shape[0], var.shape[1],
              _stream())
    return g_var


def iterate_update(x, g_opt):
    """g_opt += nan_to_num(x) - g_opt (styler_2p.py:259-262) in place, and x <- the new g_opt"""
    assert x.shape == g_opt.shape
    _lib.call("nfs_iterate_update", _ptr(x), _ptr(g_opt), x.numel(), _stream())
    _written(x, g_opt)


def fill(x, value):
    _lib.call("nfs_fill", _ptr(x), float(value), x.numel(), _stream())


def axpy(y, x, a):
    _lib.call("nfs_axpy", _ptr(y), _ptr(x), float(a), x.numel(), _stream())


# ---- SURVEY 8(f)-1 -------------------------------------------------------------------

def g2p_fwd(g, p, cubic=True):
    """g [X,Y,(Z),C] cell-centred grid, p [N,nd] in [0,1] -> [N,C]  (transform.py:771-1231, forward only)"""
    nd = p.shape[-1]
    assert g.dim() == nd + 1 and nd in (2, 3)
    N, Cn = p.shape[0], g.shape[-1]
    out = _empty((N, Cn), g)
    X, Y = g.shape[0], g.shape[1]
    Z = g.shape[2] if nd == 3 else 1
    _lib.call("nfs_g2p_fwd", _ptr(g), _ptr(p), _ptr(out), nd, X, Y, Z, Cn, N, int(bool(cubic)), _stream())
    return out



# ---- node types of the Inception-v1 loss network ------------------------------------------------------
# Operands are channel RANGES of contiguous [B,H,W,ld] buffers: (tensor, first channel, channels).

def _rows(t):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 4):
        raise ValueError("expected a contiguous float32 CUDA tensor [B,H,W,ld]")
    return t


def same_out(n, k, stride):
    """TF SAME: (output size, padding before)"""
    out = -(-n // stride)
    return out, max((out - 1) * stride + k - n, 0) // 2


def conv2d_pack(w_hwio, transpose=False):
    """HWIO filters -> the layout ``conv2d_fwd`` reads; ``transpose``: the data-gradient filters of a stride-1 conv"""
    w = w_hwio.contiguous()
    kh, kw, Ci, Co = w.shape
    n = _lib.lib().nfs_conv2d_packed_floats(kh, kw, Ci, Co, int(transpose))
    packed = _empty((n,), w)
    _lib.call("nfs_conv2d_pack", _ptr(w), _ptr(packed), kh, kw, Ci, Co, int(transpose), _stream())
    return packed


def conv2d_fwd(x, cx, Cin, packed, bias, y, cy, Cout, kh, kw, stride=1, relu=True, y_pre=None, cp=0, x_mask=None, cm=0,
               accumulate=False):
    """y[..., cy:cy+Cout] (+)= relu(conv_SAME(x[..., cx:cx+Cin] * (x_mask[..., cm:cm+Cin] > 0)) + bias);
    y_pre[..., cp:cp+Cout] = the value before the ReLU"""
    _rows(x), _rows(y)
    B, H, W, ldx = x.shape
    Ho, _ = same_out(H, kh, stride)
    Wo, _ = same_out(W, kw, stride)
    assert tuple(y.shape[:3]) == (B, Ho, Wo), (tuple(y.shape), (B, Ho, Wo))
    assert cx + Cin <= ldx and cy + Cout <= y.shape[3]
    xm = None if x_mask is None else _rows(x_mask).data_ptr() + 4 * cm
    yp = None if y_pre is None else _rows(y_pre).data_ptr() + 4 * cp
    if x_mask is not None:
        assert tuple(x_mask.shape[:3]) == (B, H, W) and cm + Cin <= x_mask.shape[3]
    if y_pre is not None:
        assert tuple(y_pre.shape[:3]) == (B, Ho, Wo) and cp + Cout <= y_pre.shape[3]
    nws = _lib.lib().nfs_conv2d_workspace_floats(B, H, W, Cin, Cout, kh, kw, stride)
    ws = conv_workspace(x.device, nws) if nws > 0 else None     # shared scratch (calls on one stream are ordered)
    _lib.call("nfs_conv2d_fwd", x.data_ptr() + 4 * cx, ldx, xm, 0 if x_mask is None else x_mask.shape[3], _ptr(packed),
              _ptr(bias), y.data_ptr() + 4 * cy, y.shape[3], yp, 0 if y_pre is None else y_pre.shape[3], B, H, W, Cin,
              Cout, kh, kw, stride, int(relu), int(accumulate), _ptr(ws), 0 if ws is None else ws.numel(), _stream())
    return y


_group_ws = {}


def conv2d_group(problems):
    """Several stride-1 SAME convolutions of one batch in one launch.  ``problems``: dicts with the arguments of
    ``conv2d_fwd`` (x, cx, Cin, packed, bias, y, cy, Cout, k, relu, y_pre, cp, x_mask, cm, accumulate) and
    ``sum_with_prev`` (added to the previous problem's result: same pixels, Cout and output range)"""
    import ctypes
    n = len(problems)
    arr = (_lib.ConvDesc * n)()
    B = problems[0]["x"].shape[0]
    for q, pr in zip(arr, problems):
        x, y = _rows(pr["x"]), _rows(pr["y"])
        assert x.shape[0] == B and tuple(y.shape[:3]) == tuple(x.shape[:3])
        cx, cy, Cin, Cout = pr.get("cx", 0), pr.get("cy", 0), pr["Cin"], pr["Cout"]
        assert cx + Cin <= x.shape[3] and cy + Cout <= y.shape[3]
        xm, yp = pr.get("x_mask"), pr.get("y_pre")
        q.x, q.ldx = x.data_ptr() + 4 * cx, x.shape[3]
        q.y, q.ldy = y.data_ptr() + 4 * cy, y.shape[3]
        q.x_mask, q.ldm = (None, 0) if xm is None else (_rows(xm).data_ptr() + 4 * pr.get("cm", 0), xm.shape[3])
        q.y_pre, q.ldp = (None, 0) if yp is None else (_rows(yp).data_ptr() + 4 * pr.get("cp", 0), yp.shape[3])
        q.packed, q.bias = _ptr(pr["packed"]), _ptr(pr.get("bias"))
        q.H, q.W, q.Cin, q.Cout, q.kh, q.kw = x.shape[1], x.shape[2], Cin, Cout, pr["k"], pr["k"]
        q.relu, q.accumulate = int(bool(pr.get("relu", True))), int(bool(pr.get("accumulate", False)))
        q.sum_with_prev = int(bool(pr.get("sum_with_prev", False)))
    ap = ctypes.cast(arr, ctypes.c_void_p)
    key = (B,) + tuple((q.H, q.W, q.Cin, q.Cout, q.kh, q.sum_with_prev) for q in arr)
    nws = _group_ws.get(key)
    if nws is None:
        nws = _lib.lib().nfs_conv2d_group_workspace_floats(ap, n, B)
        if nws < 0:
            _lib.call("nfs_conv2d_group", ap, n, B, None, 0, _stream())     # raises with the planner's message
        _group_ws[key] = nws
    ws = conv_workspace(x.device, nws) if nws > 0 else None
    if _lib.PROFILE is not None:                          # (bench: the shapes behind this call's event pair)
        _lib.PROFILE.setdefault("shapes:nfs_conv2d_group", []).append(
            [(B, q.H, q.W, q.Cin, q.Cout, q.kh, q.kw) for q in arr])
    _lib.call("nfs_conv2d_group", ap, n, B, _ptr(ws), 0 if ws is None else ws.numel(), _stream())


def conv2d_dgrad_small(gy, cg, Co, w_hwio, in_hw, stride, y_act=None, ca=0):
    """data gradient of a convolution with <= 4 input channels down to the image [B,H,W,Ci]"""
    _rows(gy)
    kh, kw, Ci, Co_w = w_hwio.shape
    assert Co_w == Co
    B = gy.shape[0]
    H, W = in_hw
    assert tuple(gy.shape[1:3]) == (same_out(H, kh, stride)[0], same_out(W, kw, stride)[0])
    gx = _empty((B, H, W, Ci), gy)
    ya = None if y_act is None else _rows(y_act).data_ptr() + 4 * ca
    _lib.call("nfs_conv2d_dgrad_small", gy.data_ptr() + 4 * cg, gy.shape[3], ya, 0 if y_act is None else y_act.shape[3],
              _ptr(w_hwio.contiguous()), _ptr(gx), B, H, W, Ci, Co, kh, kw, stride, _stream())
    return gx


def maxpool3_fwd(x, stride):
    """3x3 SAME max pool over every float of the rows: (y [B,Ho,Wo,ld], arg uint8 [B,Ho,Wo,ld])"""
    _rows(x)
    B, H, W, ld = x.shape
    Ho, Wo = same_out(H, 3, stride)[0], same_out(W, 3, stride)[0]
    y = _empty((B, Ho, Wo, ld), x)
    arg = torch.empty((B, Ho, Wo, ld), dtype=torch.uint8, device=x.device)
    _lib.call("nfs_maxpool3_fwd", _ptr(x), _ptr(y), arg.data_ptr(), B, H, W, ld, stride, _stream())
    return y, arg


def maxpool3_bwd(gy, arg, in_hw, stride, gx=None, relu_of=None):
    """gx (+)= the pool adjoint; ``gx`` given: accumulate into it; ``relu_of`` (the pooled tensor, a ReLU output): the
    result also carries that ReLU's adjoint"""
    _rows(gy)
    B, Ho, Wo, ld = gy.shape
    H, W = in_hw
    acc = gx is not None
    if gx is None:
        gx = _empty((B, H, W, ld), gy)
    assert tuple(gx.shape) == (B, H, W, ld) and arg.dtype == torch.uint8 and arg.is_contiguous()
    assert relu_of is None or tuple(relu_of.shape) == tuple(gx.shape)
    _lib.call("nfs_maxpool3_bwd", _ptr(gy), arg.data_ptr(), _ptr(gx), B, H, W, ld, stride, int(acc), _ptr(relu_of),
              _stream())
    return gx


def lrn_fwd(x, C, radius, bias, alpha, beta):
    """tf.nn.lrn over the first C channels of the rows: (y, scale), both shaped like x (padding channels zero)"""
    _rows(x)
    ld = x.shape[3]
    y = _zeros(x.shape, x) if ld > C else _empty(x.shape, x)
    scale = _empty(x.shape, x)
    _lib.call("nfs_lrn_fwd", _ptr(x), _ptr(y), _ptr(scale), x.numel() // ld, C, ld, int(radius), float(bias),
              float(alpha), float(beta), _stream())
    return y, scale


def lrn_bwd(x, y, scale, gy, C, radius, alpha, beta, gx=None):
    _rows(gy)
    ld = x.shape[3]
    acc = gx is not None
    if gx is None:
        gx = _zeros(x.shape, x) if ld > C else _empty(x.shape, x)
    _lib.call("nfs_lrn_bwd", _ptr(x), _ptr(y), _ptr(scale), _ptr(gy), _ptr(gx), x.numel() // ld, C, ld, int(radius),
              float(alpha), float(beta), int(acc), _stream())
    return gx


def avgpool_valid_fwd(x, k):
    """k x k average pool, stride 1, VALID, over every float of the rows"""
    _rows(x)
    B, H, W, ld = x.shape
    y = _empty((B, H - k + 1, W - k + 1, ld), x)
    _lib.call("nfs_avgpool_valid_fwd", _ptr(x), _ptr(y), B, H, W, ld, int(k), _stream())
    return y


def avgpool_valid_bwd(gy, in_hw, k, gx=None):
    _rows(gy)
    B, _, _, ld = gy.shape
    H, W = in_hw
    acc = gx is not None
    if gx is None:
        gx = _empty((B, H, W, ld), gy)
    _lib.call("nfs_avgpool_valid_bwd", _ptr(gy), _ptr(gx), B, H, W, ld, int(k), int(acc), _stream())
    return gx


def relu_mask_add(g, cg, act, ca, addend, cadd, C, out=None, co=0):
    """out[..., co:co+C] = g[..., cg:cg+C] * (act[..., ca:ca+C] > 0) + addend[..., cadd:cadd+C]; g / act / addend may
    be None"""
    ref = g if g is not None else addend
    _rows(ref)
    npix = ref.numel() // ref.shape[3]
    if out is None:
        out = _empty(tuple(ref.shape[:3]) + (C,), ref)
        co = 0

    def at(t, c):
        return (None, 0) if t is None else (_rows(t).data_ptr() + 4 * c, t.shape[3])
    (pg, lg), (pa, la), (pd, ldd) = at(g, cg), at(act, ca), at(addend, cadd)
    _lib.call("nfs_relu_mask_add", pg, lg, pa, la, pd, ldd, out.data_ptr() + 4 * co, out.shape[3], npix, C, _stream())
    return out
