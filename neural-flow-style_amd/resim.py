"""the particle resampler that produces the particle sets the smokegun stylisation
consumes -- counterpart of ``SimG2P`` in the reference's test_smokegun_resim.py:17-217, on the HIP
operators of this package (g2p, p2g, p2g_wavg, TF-Adam).  Same method names, argument meaning and result
keys; numpy in, numpy out.

    advect:    RK4 sampling of the cell-centred velocity field at the particles (cubic g2p), step 0.5
    optimize:  advect -> ``iter`` TF-Adam steps on a particle displacement against the pressure loss
               mean(where(d_rec > 0, d_rec - rho0, 0)^2) of the SPH splat -> seed new particles where the
               splatted density misses the target -> multi-scale density sampling at the particles
"""
import numpy as np
import torch

from . import ops
from . import transform as T
from .engine import TFAdamState


def mac_to_centered(v_):
    """mantaflow MAC-grid velocity [D,H,W,3] -> cell-centred, H flipped (test_smokegun_resim.py:233-243)"""
    v_ = np.asarray(v_)
    vx = np.dstack((v_, np.zeros((v_.shape[0], v_.shape[1], 1, v_.shape[3]), v_.dtype)))
    vx = (vx[:, :, 1:, 0] + vx[:, :, :-1, 0]) * 0.5
    vy = np.hstack((v_, np.zeros((v_.shape[0], 1, v_.shape[2], v_.shape[3]), v_.dtype)))
    vy = (vy[:, 1:, :, 1] + vy[:, :-1, :, 1]) * 0.5
    vz = np.vstack((v_, np.zeros((1, v_.shape[1], v_.shape[2], v_.shape[3]), v_.dtype)))
    vz = (vz[1:, :, :, 2] + vz[:-1, :, :, 2]) * 0.5
    return np.stack([vx, vy, vz], axis=-1)[:, ::-1]


def velocity_to_normalised(v_, scale):
    """cell-centred (x,y,z) velocity in cells/frame -> (z,y,x) components in [0,1] domain units, y up
    (test_smokegun_resim.py:245-248)"""
    vx = v_[..., 0] / v_.shape[2] * scale
    vy = -v_[..., 1] / v_.shape[1] * scale
    vz = v_[..., 2] / v_.shape[0] * scale
    return np.stack([vz, vy, vx], axis=-1).astype(np.float32)


class SimG2P(object):
    def __init__(self, self_dict, device="cuda"):
        for arg in vars(self_dict):
            setattr(self, arg, getattr(self_dict, arg))
        self.device = torch.device(device)
        self.src_region = getattr(self, "src_region", None)   # ((z0,z1),(y0,y1),(x0,x1)) seeding box

    # -- helpers ---------------------------------------------------------------------------
    def _dev(self, a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=self.device)

    def _advect(self, x, u):
        """x [N,3] device, u [D,H,W,3] device -> x + 0.5 * RK4 velocity (test_smokegun_resim.py:35-53)"""
        xb, ub = x.unsqueeze(0), u.unsqueeze(0)
        v = T.g2p(ub, xb, is_2d=False)
        v1 = T.g2p(ub, xb + v * 0.5, is_2d=False)
        v2 = T.g2p(ub, xb + v1 * 0.5, is_2d=False)
        v3 = T.g2p(ub, xb + v2, is_2d=False)
        v = (v + v1 * 2 + v2 * 2 + v3) / 6
        return (xb + v * 0.5)[0]

    def _pressure_loss(self, x_hat):
        d_rec = T.p2g(x_hat.unsqueeze(0), self.domain, self.resolution, self.radius, self.rest_density, self.nsize,
                      kernel="cubic", support=4, clip=False, is_2d=False)
        pres = torch.where(d_rec > 0, d_rec - self.rest_density, torch.zeros_like(d_rec))
        return (pres ** 2).mean()

    def _density_sampling(self, x_hat, d):
        """multi-scale particle density sampling (test_smokegun_resim.py:83-106)"""
        dd = d.unsqueeze(0).unsqueeze(-1)
        xb = x_hat.unsqueeze(0)
        r = []
        d_hat = None
        for o in range(self.octave_n):
            if o > 0:
                d_hi = d_hat
                d_ = (dd - d_hi.flip(2)).contiguous()
            else:
                d_ = dd
            r_ = T.g2p(d_, xb, is_2d=False)
            r.append(r_)
            factor = self.octave_scale ** o
            d_hat = T.p2g_wavg(xb, r_, self.domain, self.resolution, self.radius, self.nsize, kernel="cubic",
                               is_2d=False, clip=False, support=self.support / factor)
            if o > 0:
                d_hat = d_hat + d_hi
        r_smp = torch.cat(r, dim=-1)[0]
        d_smp = d_hat[0, ..., 0].clamp(0, 1)
        d_diff = (dd.flip(2) - d_hat)[0].flip(1)[..., 0]
        return r_smp, d_smp, d_diff

    # -- reference surface -----------------------------------------------------------------
    def sample(self, d, disc=1, threshold=0, p0=None, p_id=None):
        """seed ``disc^3`` particles per cell where d > threshold inside the source region (the reference
        hard-codes the smokegun inflow box d[76:124,231:279,16:64], test_smokegun_resim.py:121-123; here
        ``src_region`` if set, else the whole grid); positions normalised to [0,1], (z,y,x)."""
        d = np.asarray(d)
        reg = self.src_region or ((0, d.shape[0]), (0, d.shape[1]), (0, d.shape[2]))
        sub = d[reg[0][0]:reg[0][1], reg[1][0]:reg[1][1], reg[2][0]:reg[2][1]]
        pid = np.array(np.where(sub > threshold)).transpose([1, 0]).astype(np.float64)
        pid += np.array([reg[0][0], reg[1][0], reg[2][0]])
        cell_size = 1 / disc
        offset = cell_size / 2
        p = []
        for i in range(disc):
            for j in range(disc):
                for k in range(disc):
                    p.append(pid + offset + np.array([cell_size * i, cell_size * j, cell_size * k]))
        p = np.concatenate(p, axis=0)
        p = np.stack([p[:, 0] / d.shape[0], p[:, 1] / d.shape[1], p[:, 2] / d.shape[2]], axis=-1)
        if len(p) > 0:
            if p_id is None:
                p_id = np.arange(p.shape[0])
            else:
                p_id0 = p_id[-1] + 1
                p_id = np.concatenate([p_id, np.arange(p_id0, p_id0 + p.shape[0])])
            if p0 is not None:
                p = np.concatenate([p0, p], axis=0)
        elif p0 is not None:
            p = np.asarray(p0)
        return p, p_id

    def naive_adv(self, p, u, r):
        """advect the particles and reconstruct the density from their attribute r (109-111, 155-166)"""
        x_adv = self._advect(self._dev(p), self._dev(u))
        d_rec = T.p2g_wavg(x_adv.unsqueeze(0), self._dev(r).unsqueeze(0), self.domain, self.resolution, self.radius,
                           self.nsize, kernel="cubic", is_2d=False, clip=False, support=4)
        return x_adv.cpu().numpy(), d_rec[0, ..., 0].cpu().numpy()

    def optimize(self, p, p_id, d, u):
        """1. advect p_t with u_t, optimise a displacement for redistribution (pressure loss);
        2. seed new particles where the particles do not cover the density (source region);
        3. sample the particle densities from d_(t+1).  (test_smokegun_resim.py:168-217)"""
        d_t = self._dev(d)
        p_adv = self._advect(self._dev(p), self._dev(u))
        v = torch.zeros_like(p_adv)
        adam = TFAdamState()
        losses = []
        for _ in range(self.iter):
            vv = v.clone().requires_grad_()
            loss = self._pressure_loss(p_adv + vv)
            (g,) = torch.autograd.grad(loss, vv)
            losses.append(float(loss.detach()))
            adam.step(v, g.contiguous(), self.lr)
        p_new = p_adv + v
        _, _, d_diff = self._density_sampling(p_new, d_t)
        d_diff_np = d_diff.cpu().numpy()
        p_all, p_id = self.sample(d_diff_np, disc=self.disc, threshold=self.threshold, p0=p_new.cpu().numpy(),
                                  p_id=p_id)
        x_all = self._dev(p_all)
        r_smp, d_smp, _ = self._density_sampling(x_all, d_t)
        return {"p": p_all, "p_id": p_id, "p_den": r_smp.cpu().numpy(), "l": losses,
                "d_diff": np.mean(d_diff_np, axis=0), "d_smp": d_smp.cpu().numpy(),
                "p_adv": p_adv.cpu().numpy(), "p_new": p_new.cpu().numpy()}
