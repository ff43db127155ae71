"""Host-side mirror of the reference's ``transform.py`` operator interface.

Same function names / argument meaning as the reference module, but the tensors
are PyTorch CUDA tensors and every differentiable operator is a
``torch.autograd.Function`` whose forward and backward call the HIP kernels
through the C ABI (``ops``).  View sampling (``rot_mat*``, ``PoissonDisc``) is
host NumPy and consumes the seeded RNG in the same order as the reference
(transform.py:14-150, 640-768) so a given ``config.seed`` yields the same views.
"""
from __future__ import annotations

import math

import os

import numpy as np
import torch

from . import ops

# --------------------------------------------------------------------------------------
# view sampling (host)
# --------------------------------------------------------------------------------------

def _rot(deg):
    a = deg / 180.0 * np.pi
    return np.cos(a), np.sin(a)


def rot_z_3d(deg):
    """rotation that fixes array axis 2 (W): elevation (transform.py:640-648)"""
    c, s = _rot(deg)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def rot_y_3d(deg):
    """rotation that fixes array axis 1 (H): azimuth (transform.py:650-658)"""
    c, s = _rot(deg)
    return np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]])


def rot_x_3d(deg):
    c, s = _rot(deg)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


class PoissonDisc(object):
    """Bridson Poisson-disc sampling on [0,width]x[0,height] with minimum
    distance r; same draw order as transform.py:14-150 (start point, then
    rng.choice over the active list, then (rho, theta) candidates, k=30)."""

    _NEIGH = [(dx, dy) for dy in range(-2, 3) for dx in range(-2, 3) if abs(dx) + abs(dy) < 4]

    def __init__(self, rng, width=50, height=50, r=1, k=30):
        self.rng, self.width, self.height, self.r, self.k = rng, width, height, r, k
        self.a = r / np.sqrt(2)
        self.nx, self.ny = int(width / self.a) + 1, int(height / self.a) + 1
        self.cells = {}
        self.samples = []

    def _cell(self, pt):
        return int(pt[0] // self.a), int(pt[1] // self.a)

    def _far_enough(self, pt):
        cx, cy = self._cell(pt)
        for dx, dy in self._NEIGH:
            j = self.cells.get((cx + dx, cy + dy))
            if j is None:
                continue
            q = self.samples[j]
            if (q[0] - pt[0]) ** 2 + (q[1] - pt[1]) ** 2 < self.r ** 2:
                return False
        return True

    def _candidate(self, ref):
        tries = 0
        while tries < self.k:
            rho = self.rng.uniform(self.r, 2 * self.r)
            th = self.rng.uniform(0, 2 * np.pi)
            pt = (ref[0] + rho * np.cos(th), ref[1] + rho * np.sin(th))
            if not (0 < pt[0] < self.width and 0 < pt[1] < self.height):
                continue  # outside draws are not counted (transform.py:109-111)
            if self._far_enough(pt):
                return pt
            tries += 1
        return None

    def sample(self):
        first = (self.rng.uniform(0, self.width), self.rng.uniform(0, self.height))
        self.samples = [first]
        self.cells = {self._cell(first): 0}
        active = [0]
        while active:
            j = self.rng.choice(active)
            pt = self._candidate(self.samples[j])
            if pt is None:
                active.remove(j)
                continue
            self.samples.append(pt)
            active.append(len(self.samples) - 1)
            self.cells[self._cell(pt)] = len(self.samples) - 1
        return self.samples


def rot_mat_uniform(phi0, phi1, phi_unit, theta0, theta1, theta_unit):
    """lattice of views (transform.py:750-768; int() added for modern NumPy)"""
    def axis(a0, a1, unit):
        if unit == 0:
            return [(a1 - a0) / 2]
        return np.linspace(a0, a1, int(np.abs(a1 - a0) / float(unit) + 1), endpoint=True)
    return [{"phi": p, "theta": t} for p in axis(phi0, phi1, phi_unit) for t in axis(theta0, theta1, theta_unit)]


def rot_mat_poisson(phi0, phi1, phi_unit, theta0, theta1, theta_unit, rng):
    """Poisson-disc views in the (theta, phi) box (transform.py:722-748)"""
    if phi_unit == 0:
        h, phi0 = 1, -0.5
    else:
        h = phi1 - phi0
    if theta_unit == 0:
        w, theta0 = 1, -0.5
    else:
        w = theta1 - theta0
    r = max(phi_unit, theta_unit) / 2
    pts = PoissonDisc(rng, height=h, width=w, r=r).sample()
    return [{"phi": s[1] + phi0, "theta": s[0] + theta0} for s in pts]


def rot_mat(phi0, phi1, phi_unit, theta0, theta1, theta_unit, sample_type="uniform", rng=None, nv=None):
    """-> (list of 3x3 R = Ry(theta) @ Rz(phi), list of views)   (transform.py:683-720)"""
    def fit(views, unit_scale):
        if nv is None:
            return views
        if len(views) > nv:
            return views[len(views) - nv:]
        if len(views) < nv:
            extra = rot_mat_poisson(phi0, phi1, phi_unit * unit_scale, theta0, theta1, theta_unit * unit_scale, rng)
            views = views + extra[:nv - len(views)]
        return views

    if "uniform" in sample_type:
        views = rot_mat_uniform(phi0, phi1, phi_unit, theta0, theta1, theta_unit)
    elif "poisson" in sample_type:
        views = rot_mat_poisson(phi0, phi1, phi_unit, theta0, theta1, theta_unit, rng)
        views += rot_mat_uniform(phi0, phi1, 0, theta0, theta1, 0)
        views = fit(views, 1)
    else:  # both
        views = rot_mat_uniform(phi0, phi1, phi_unit, theta0, theta1, theta_unit)
        views += rot_mat_poisson(phi0, phi1, phi_unit * 2, theta0, theta1, theta_unit * 2, rng)
        views = fit(views, 2)
    mats = [np.matmul(rot_y_3d(v["theta"]), rot_z_3d(v["phi"])) for v in views]
    return mats, views


def rot_to_device(mats, device):
    return torch.tensor(np.asarray(mats, dtype=np.float32).reshape(-1, 3, 3), device=device)


# --------------------------------------------------------------------------------------
# differentiable operators (HIP)
# --------------------------------------------------------------------------------------

class _Warp3d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, imgs, coords):
        imgs = imgs.contiguous(); coords = coords.contiguous()
        ctx.save_for_backward(imgs, coords)
        return ops.warp3d_fwd(imgs, coords)

    @staticmethod
    def backward(ctx, g):
        imgs, coords = ctx.saved_tensors
        gi, gc = ops.warp3d_bwd(imgs, coords, g.contiguous(), need_coords=ctx.needs_input_grad[1])
        return gi, gc


def batch_warp3d(imgs, mappings, sample_shape=None):
    """imgs [B,X,Y,Z,C], mappings [B,3,X,Y,Z] -> [B,X,Y,Z,C]   (transform.py:238-269)"""
    return _Warp3d.apply(imgs, mappings)


class _Rotate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d, rot):
        ctx.save_for_backward(rot)
        ctx.shape = d.shape
        return ops.rotate_fwd(d.contiguous().reshape(d.shape[-4:]), rot)

    @staticmethod
    def backward(ctx, g):
        (rot,) = ctx.saved_tensors
        return ops.rotate_bwd(g.contiguous(), rot).reshape(ctx.shape), None


def rotate(d, rot_mat_):
    """d [1,D,H,W,C], rot_mat_ [V,3,3] (device tensor; the reference feeds it through a
    placeholder, transform.py:617) -> [V,D,H,W,C]"""
    assert d.shape[0] == 1, "one volume per call (the reference tiles it per view)"
    return _Rotate.apply(d, rot_mat_.contiguous())


class _Advect(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d, vel):
        d = d.contiguous(); vel = vel.contiguous()
        ctx.save_for_backward(d, vel)
        return ops.advect_fwd(d[0], vel[0]).unsqueeze(0)

    @staticmethod
    def backward(ctx, g):
        d, vel = ctx.saved_tensors
        gd, gv = ops.advect_bwd(d[0], vel[0], g.contiguous()[0], need_d=ctx.needs_input_grad[0],
                                need_vel=ctx.needs_input_grad[1])
        return (None if gd is None else gd.unsqueeze(0)), (None if gv is None else gv.unsqueeze(0))


class _Advect2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d, vel):
        d = d.contiguous(); vel = vel.contiguous()
        ctx.save_for_backward(d, vel)
        return ops.advect2d_fwd(d[0], vel[0]).unsqueeze(0)

    @staticmethod
    def backward(ctx, g):
        d, vel = ctx.saved_tensors
        gd, gv = ops.advect2d_bwd(d[0], vel[0], g.contiguous()[0], need_d=ctx.needs_input_grad[0],
                                  need_vel=ctx.needs_input_grad[1])
        return (None if gd is None else gd.unsqueeze(0)), (None if gv is None else gv.unsqueeze(0))


def advect(d, vel, order=1, is_3d=False):
    """semi-Lagrangian step d(x - v) (transform.py:557-609): 3-D d [1,D,H,W,C] + vel [1,D,H,W,3], or 2-D d [1,H,W,C]
    + vel [1,H,W,2] (``is_3d=False``, the reference's default).  order 1 is differentiable in d and vel; order 2 is
    the MacCormack scheme with its extrema limiter done as intended (the reference's limiter lines do not run,
    transform.py:577,598-601), forward only."""
    assert d.shape[0] == 1
    is_3d = bool(is_3d) or d.dim() == 5
    if order == 1:
        return _Advect.apply(d, vel) if is_3d else _Advect2d.apply(d, vel)
    return ops.advect_maccormack(d.detach()[0].contiguous(), vel.detach()[0].contiguous()).unsqueeze(0)


class _Warp2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, imgs, coords):
        imgs = imgs.contiguous(); coords = coords.contiguous()
        ctx.save_for_backward(imgs, coords)
        return ops.warp2d_fwd(imgs, coords)

    @staticmethod
    def backward(ctx, g):
        imgs, coords = ctx.saved_tensors
        gi, gc = ops.warp2d_bwd(imgs, coords, g.contiguous(), need_imgs=ctx.needs_input_grad[0],
                                need_coords=ctx.needs_input_grad[1])
        return gi, gc


def batch_warp2d(imgs, mappings, sample_shape=None):
    """imgs [B,X,Y,C], mappings [B,2,X,Y] -> [B,X,Y,C]   (transform.py:206-236)"""
    return _Warp2d.apply(imgs, mappings)


class _Curl(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s):
        return ops.curl_fwd(s.contiguous())

    @staticmethod
    def backward(ctx, g):
        return ops.curl_bwd(g.contiguous())


def curl(s, is_2d=True):
    """velocity of a stream function (transform.py:517-555): 2-D s [B,H,W,1] -> [B,H,W,2] = (ds/dy, -ds/dx);
    3-D s [B,D,H,W,3] -> [B,D,H,W,3].  Differentiable (the TNST parametrisation optimises s)."""
    if is_2d:
        return torch.stack([_Curl.apply(s[b, ..., 0]) for b in range(s.shape[0])])
    return torch.stack([_Curl.apply(s[b]) for b in range(s.shape[0])])


class _Permute(torch.autograd.Function):
    """x[:, order] for a PERMUTATION ``order`` of axis 1.  The adjoint of a gather through a permutation is a plain scatter
    (every target written exactly once: ``index_copy_``, no accumulation) -- autograd's generic index backward (a sorted
    scatter-add) takes 137 us for 5e5 x 3 floats, this 8"""

    @staticmethod
    def forward(ctx, x, order):
        ctx.save_for_backward(order)
        return x.index_select(1, order)

    @staticmethod
    def backward(ctx, g):
        (order,) = ctx.saved_tensors
        return torch.empty_like(g).index_copy_(1, order, g), None


def permute_particles(x, order):
    """x [1,N,k] seen through the permutation ``order`` (see grid_order)"""
    return _Permute.apply(x, order)


def grid_order(p, resolution, brick=8, stable=True):
    """Permutation that puts particles p [N,nd] (in [0,1], axis order = array order) in the order of the grid, brick by
    brick (``brick``^nd cells) and cell by cell inside a brick.  The splat accumulates a block of consecutive particles
    in LDS when their cells sit in a small box (csrc/splat.hip): bricks keep that box small in every direction, and
    keep it small while a Lagrangian run moves the particles by a few cells (a plain row-major cell order puts a
    particle that drifts one plane up 40 000 cells away in the 200^3 grid)."""
    nd = p.shape[1]
    dims = torch.tensor([float(v) for v in resolution[:nd]], device=p.device)
    cell = torch.minimum((p.clamp(min=0) * dims).floor(), dims - 1).long()
    nb = [(int(v) + brick - 1) // brick for v in resolution[:nd]]
    bk, ck = torch.zeros_like(cell[:, 0]), torch.zeros_like(cell[:, 0])
    for k in range(nd):
        bk = bk * nb[k] + cell[:, k] // brick
        ck = ck * brick + cell[:, k] % brick
    key = bk * (brick ** nd) + ck
    if int(np.prod(nb)) * brick ** nd < 2 ** 31:
        key = key.to(torch.int32)       # (32-bit keys take the radix sort: ~10x faster than the 64-bit merge sort at 5e5 keys)
    # stable: particles of one cell keep their relative order (reproducible layouts: run to run and rank to rank)
    return torch.argsort(key, stable=stable)


class _P2G(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, pc, pd, cfg):
        ctx.cfg = cfg
        p2 = p[0].contiguous()
        a = None if pc is None else pc[0].contiguous()
        r = None if pd is None else pd[0].reshape(-1).contiguous()
        ctx.save_for_backward(p2, a, r)
        ctx.has = (pc is not None, pd is not None)
        return ops.p2g_fwd(p2, cfg, attr=a, pd=r).unsqueeze(0)

    @staticmethod
    def backward(ctx, g):
        p2, a, r = ctx.saved_tensors
        need_p, need_a, need_r = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        gp, ga, gr = ops.p2g_bwd(p2, ctx.cfg, g.contiguous()[0], attr=a, pd=r, need_p=need_p,
                                 need_attr=need_a and a is not None, need_pd=need_r and r is not None)
        return (None if gp is None else gp.unsqueeze(0), None if ga is None else ga.unsqueeze(0),
                None if gr is None else gr.reshape(1, -1, 1), None)


def p2g(p, domain, res, radius, rest_density, nsize, pc=None, pd=None, is_2d=True, kernel="cubic", eps=1e-6,
        clip=True, support=4):
    """SPH splat, p [1,N,d] in [0,1] -> [1,*res,1|C]  (transform.py:1310-1453)"""
    assert kernel == "cubic", "the stylers only use the cubic spline kernel"
    nd = 2 if is_2d else 3
    cfg = ops.make_splat_cfg(nd, list(res), list(domain), radius, support, rest_density, nsize, clip,
                             0 if pc is None else 1)
    return _P2G.apply(p, pc, pd, cfg)


class _P2GWavg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, x, cfg, eps):
        p2 = p[0].contiguous(); a = x[0].contiguous()
        xs, ws = ops.p2g_fwd(p2, cfg, attr=a)
        ctx.cfg, ctx.eps = cfg, eps
        ctx.save_for_backward(p2, a, xs, ws)
        return ops.p2g_wavg_finish(xs, ws, eps).unsqueeze(0)

    @staticmethod
    def backward(ctx, g):
        p2, a, xs, ws = ctx.saved_tensors
        fused = None
        if os.environ.get("NFS_SPLAT_LDS", "1") != "0":
            fused = ops.p2g_wavg_bwd(p2, ctx.cfg, xs, ws, g.contiguous()[0], a, ctx.eps,
                                     need_p=ctx.needs_input_grad[0], need_attr=ctx.needs_input_grad[1])
        if fused is not None:
            gp, ga = fused
        else:
            g_xs, g_ws = ops.p2g_wavg_finish_bwd(xs, ws, g.contiguous()[0], ctx.eps)
            gp, ga, _ = ops.p2g_bwd(p2, ctx.cfg, g_xs, attr=a, g_wsum=g_ws, need_p=ctx.needs_input_grad[0],
                                    need_attr=ctx.needs_input_grad[1])
        return (None if gp is None else gp.unsqueeze(0), None if ga is None else ga.unsqueeze(0), None, None)


def p2g_wavg(p, x, domain, res, radius, nsize, is_2d=True, kernel="cubic", eps=1e-6, clip=True, support=4):
    """weighted-average splat (transform.py:1577-1704), cubic kernel as styler_3p.py:83 uses it"""
    assert kernel == "cubic"
    nd = 2 if is_2d else 3
    cfg = ops.make_splat_cfg(nd, list(res), list(domain), radius, support, 1.0, nsize, clip, 2)
    return _P2GWavg.apply(p, x, cfg, eps)


def g2p(g, p, is_2d=True, is_linear=False):
    """grid -> particle sampling, g [1,X,Y,(Z),C] cell-centred, p [1,N,d] in [0,1] -> [1,N,C]
    (transform.py:771-776: Catmull-Rom cubic unless ``is_linear``).  Forward only, like its only
    consumer in the reference (the SimG2P resampler, test_smokegun_resim.py:36-47,96)."""
    assert g.shape[0] == 1 and p.shape[0] == 1
    return ops.g2p_fwd(g[0].detach().contiguous(), p[0].detach().contiguous(), cubic=not is_linear).unsqueeze(0)


def g2p_cubic(g, p, is_2d=True):
    """the Catmull-Rom branch under its own name (transform.py:778)"""
    return g2p(g, p, is_2d=is_2d, is_linear=False)


def g2p_linear(g, p, is_2d=True):
    """the (bi/tri)linear branch under its own name (transform.py:1110)"""
    return g2p(g, p, is_2d=is_2d, is_linear=True)
