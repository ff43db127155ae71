"""Grid-sequence stylizer (BASELINE ``configs[3]``: a frame sequence on a 3-D smoke grid with semi-Lagrangian
transport of the stylised field) -- the Eulerian counterpart of ``styler_3p.Styler.run``.

The mounted reference branch keeps the loop for particles (styler_3p.py:229-439) and the grid operators as
leftovers: ``advect`` (transform.py:557-569) and ``StylerBase._transport`` (styler_base.py:59-89), which moves a grid
field from frame a to frame b through the simulation velocities.  This module assembles them exactly the way the
particle loop is built:

  per key frame t (styler_3p.py:304-363), variable re-assigned from ``g_opt[t]`` (312), one TF-Adam state per
  ``opt_id = t // frames_per_opt`` (315-323):
        d^_t = advect(d_t, v^_t)            ('v': stylisation velocity [D,H,W,3];  'd': the density itself)
        -> 3x3x3 smooth + max(.,0) -> rotate (all views, ``views=sum``) -> render -> VGG-19 -> Gram style loss
        -> adjoint chain -> one Adam step;      upd_t = new - g_opt[t]   ('d': masked by the original density, 361-363)
  temporal alignment of the updates (380-386).  Per-particle attributes ride on the particles, so the reference filters
  them along the frame axis in place (``denoise``: SciPy Gaussian, reflect, 4 sigma).  A grid field has to be carried
  to the frame it is added to:
        D_t = sum_s W[t,s] * transport(upd_s, u, s -> t)
  with W the very matrix of that Gaussian filter (util.temporal_weights; for u == 0 this IS ``denoise``) and
  ``transport`` = ``_transport`` (recursive chain of ``advect`` by +u_i forwards / -u_i backwards, or its one-step
  ``recursive=False`` form).  ``advect`` is linear in the advected field, so the sum over s is evaluated in Horner
  form -- R advects per side and frame instead of R(R+1)/2:
        S = W[t,t-R] upd_{t-R};  S = advect(S, u_{t-k}) + W[t,t-k+1] upd_{t-k+1} ...;  D_t^- = advect(S, u_{t-1})
  then ``g_opt[t] += D_t`` (386), frame interpolation (392-397), final inference.

Frames shard over ranks: contiguous blocks of optimiser groups per rank
(parallel.plan_frames), every rank keeps full view batches, and the only exchange is the halo of per-frame updates
the temporal filter reaches (point-to-point, parallel.exchange_frames) plus a handful of loss scalars.  The sharded
run reproduces the single-rank trajectory (tests: world-2 gloo, two ranks on one GPU).
"""
from __future__ import annotations

import numpy as np
import torch

from . import engine, ops, parallel
from . import transform as T
from .styler_base import StylerBase
from .util import temporal_weights


class Styler(StylerBase):
    """``Styler(config).run(params)`` with ``params['d']`` = F density frames [D,H,W] and ``params['v']`` = F
    simulation velocities [D,H,W,3] in ``advect`` units (normalised: one cell = 2/(n-1), component k along array
    axis k).  ``config.grid_variable`` = 'v' (default) or 'd'."""

    def __init__(self, self_dict):
        StylerBase.__init__(self, self_dict)
        assert self.batch_size == 1, "batch_size > 1 is not supported (styler_3p module docstring)"
        self.target = getattr(self, "grid_variable", "") or "v"
        assert self.target in ("v", "d")
        if self.rotate:
            self.rot_mat_, self.views = T.rot_mat(self.phi0, self.phi1, self.phi_unit, self.theta0, self.theta1,
                                                  self.theta_unit, sample_type=self.sample_type, rng=self.rng,
                                                  nv=self.n_views)
            if self.n_views is None:
                self.n_views = len(self.views)
        self.loss = self._make_loss(rotate=self.rotate)
        self._identity = T.rot_to_device([np.identity(3)], self.device)
        self.recursive = bool(getattr(self, "transport_recursive", True))
        self.pg = None          # set by the driver: frames shard over the ranks of this group

    # ---- helpers ---------------------------------------------------------------------------------------------------
    def _dev(self, a):
        return torch.as_tensor(np.asarray(a, np.float32)).to(self.device).contiguous()

    def _rank_world(self):
        return (0, 1) if self.pg is None else parallel.rank_world(self.pg)

    def _rot(self):
        if not self.rotate:
            return self._identity
        return T.rot_to_device(self.rot_mat_, self.device)

    def _resample_views(self):
        if self.rotate and "uniform" not in self.sample_type:
            self.rot_mat_, self.views = T.rot_mat(self.phi0, self.phi1, self.phi_unit, self.theta0, self.theta1,
                                                  self.theta_unit, sample_type=self.sample_type, rng=self.rng,
                                                  nv=self.n_views)

    def aligned_update(self, t, upd, u, W, keys):
        """D_t = sum_s W[t,s] transport(upd_s, u, s -> t) over the key frames s.  ``upd`` {frame: [D,H,W,C]} must hold
        every frame with a non-zero weight; ``u`` {frame: [D,H,W,3]} the simulation velocities.  Every frame crossing
        is one ``nfs_transport_step`` launch (advect + weighted accumulation fused); ``transport`` is
        ``StylerBase._transport`` (styler_base.py:59-74):
          recursive   a < b: g <- advect(g, +u_i), i = a..b-1;   a > b: g <- advect(g, -u_i), i = a-1..b   (Horner form)
          one-step    a < b: advect(g, +u_a (b-a));              a > b: advect(g, -u_{a-1} (a-b))          (direct sum)"""
        j = keys.index(t)
        out, w_out = upd[t], float(W[j, j])          # running result = w_out * out (scaled lazily by the next launch)
        if not self.recursive:
            for jj in range(len(keys)):
                s = keys[jj]
                if jj == j or W[j, jj] == 0.0:
                    continue
                a, scale = (s, float(t - s)) if s < t else (s - 1, -float(s - t))
                out = ops.transport_step(upd[s], u[a], scale, float(W[j, jj]), out, w_out)
                w_out = 1.0
            return out if w_out == 1.0 else out * w_out
        for sign in (+1, -1):
            # sign +1: frames s < t carried forwards; sign -1: frames s > t carried backwards
            far = [jj for jj in (range(0, j) if sign > 0 else range(len(keys) - 1, j, -1)) if W[j, jj] != 0.0]
            if not far:
                continue
            jj = far[0]
            S, w_S = upd[keys[jj]], float(W[j, jj])  # carried sum = w_S * S
            while jj != j:
                s, nxt = keys[jj], jj + sign
                s2 = keys[nxt]
                cross = list(range(s, s2)) if sign > 0 else list(range(s - 1, s2 - 1, -1))
                for n_, i in enumerate(cross):
                    add, w_add = None, 0.0
                    if n_ == len(cross) - 1:         # arriving at key frame s2: add its own (weighted) update
                        if nxt == j:
                            add, w_add = out, w_out
                        elif W[j, nxt] != 0.0:
                            add, w_add = upd[s2], float(W[j, nxt])
                    S = ops.transport_step(S, u[i], float(sign), w_S, add, w_add)
                    w_S = 1.0
                jj = nxt
            out, w_out = S, 1.0
        return out if w_out == 1.0 else out * w_out

    # ---- the optimisation loop: prepare -> iterate x iter -> finish ------------------------------------------------------
    def _frames(self, x, wanted):
        """params entry (list indexed by frame, or {frame: array}) -> {frame: device tensor} for the wanted frames"""
        if x is None:
            return None
        get = (lambda t: x[t]) if not isinstance(x, dict) else (lambda t: x.get(t))
        return {t: self._dev(get(t)) for t in wanted if get(t) is not None}

    def frames_needed(self, rank=None, world=None):
        """(density frames, simulation-velocity frames) rank ``rank`` of ``world`` touches in a sharded run: its own
        key frames, and the velocities of every frame crossing its temporal filter makes (the non-zero columns of the
        ``denoise`` matrix).  A pure function of the configuration: a driver can load exactly these frames and pass
        ``prepare(params, frames_on_device=densities)``.  A rank beyond the number of optimiser groups gets two empty
        sets (it still takes part in the collectives)."""
        if rank is None or world is None:
            rank, world = self._rank_world()
        F_ = int(self.num_frames)
        keys = list(range(0, F_, self.interp))
        plan = parallel.plan_frames(F_, self.interp, self.frames_per_opt, world)
        mine = plan[rank]
        if not mine:
            return set(), set()
        need = set(mine)
        if self.window_sigma > 0 and F_ > 1:
            Wt = temporal_weights(len(keys), self.window_sigma)
            for t in mine:
                need |= set(keys[jj] for jj in np.nonzero(Wt[keys.index(t)])[0])
            return set(mine), set(range(max(min(need) - 1, 0), min(max(need) + 1, F_)))
        return set(mine), set()

    def prepare(self, params, frames_on_device=None):
        """Put this rank's share of the sequence on the device and set up the loop state.  ``params['d']`` /
        ``params['v']`` / ``params['v_init']``: list over frames or {frame: array}; a sharded run only needs the
        frames this rank touches (``frames_needed(rank, world)``; ``frames_on_device`` = its density frames)."""
        assert self.octave_n == 1, "the grid path has one octave (the reference's octaves resize the SPLAT target, " \
                                   "styler_3p.py:241-247; a grid sequence comes at its own resolution)"
        F_ = int(self.num_frames)
        rank, world = self._rank_world()
        st = self._st = argparse_ns()
        st.F, st.rank, st.world = F_, rank, world
        st.keys = list(range(0, F_, self.interp))
        st.plan = parallel.plan_frames(F_, self.interp, self.frames_per_opt, world)
        st.owner = {t: r for r, ts in enumerate(st.plan) for t in ts}
        st.mine = st.plan[rank]
        st.Wt = temporal_weights(len(st.keys), self.window_sigma) if (self.window_sigma > 0 and F_ > 1) else None
        # which frames' updates this rank's filter reaches (non-zero weights only), and the velocities in between
        def need_of(r):
            nd = set(st.plan[r])
            if st.Wt is not None:
                for t in st.plan[r]:
                    j = st.keys.index(t)
                    nd |= set(st.keys[jj] for jj in np.nonzero(st.Wt[j])[0])
            return nd

        st.need_by_rank = [need_of(r) for r in range(world)]       # a pure function of plan and filter: no exchange
        st.need = st.need_by_rank[rank]
        want_d = set(range(F_)) if frames_on_device is None else set(frames_on_device) | set(st.mine)
        want_u = set(range(F_)) if frames_on_device is None else \
            set(range(max(min(st.need) - 1, 0), min(max(st.need) + 1, F_))) if st.need else set()
        st.d = {t: x.reshape(tuple(self.resolution)) for t, x in self._frames(params["d"], sorted(want_d)).items()}
        st.u = self._frames(params.get("v"), sorted(want_u))
        if st.Wt is not None and st.mine and not st.u:
            raise ValueError("params['v'] (simulation velocities) is needed to align the updates of a sequence")
        st.u = st.u or {}
        D, H, W_ = tuple(self.resolution)
        st.C = 3 if self.target == "v" else 1
        st.shape = (D, H, W_, st.C)
        if self.style_img is not None:
            self.loss.set_style_image(self._style_feature(self.style_img, [H, W_]))
            if getattr(self, "w_hist", 0) > 0:
                self.loss.set_hist_image(self._hist_feature(self.style_img, [H, W_]))
        if self.content_img is not None:
            self.loss.set_content_image(self._content_feature(self.content_img, [H, W_]), top_k=self._content_top_k())
        st.lr = self.lr[0] if isinstance(self.lr, list) else self.lr
        # the variable per key frame: stylisation velocity (zero, or params['v_init'][t]) / the density itself.
        # NOTE: at velocity == 0 every back-traced point sits exactly on a grid node, where the trilinear stencil has a
        # kink -- the first gradient is a one-sided derivative whose side depends on float rounding
        st.v_init = params.get("v_init")
        st.g_opt = {t: self._initial(t) for t in st.mine}
        st.work = torch.zeros(st.shape, device=self.device)     # the variable (re-assigned per frame, 312)
        # (a rank beyond the number of optimiser groups owns no frame: it only takes part in the collectives)
        first = st.d[st.mine[0]] if st.mine else (next(iter(st.d.values())) if st.d else
                                                  torch.zeros(tuple(self.resolution), device=self.device))
        st.gs = engine.GridStylizer(self.loss, first, k=self.k, target=self.target, lr=st.lr,
                                    optimizer=getattr(self, "optimizer", "adam"))
        st.opt_ = {}
        st.hist = []
        return st

    def _initial(self, t):
        st = self._st
        if self.target == "d":
            return st.d[t].reshape(st.shape).clone()
        if st.v_init is not None:
            vi = st.v_init[t] if not isinstance(st.v_init, dict) else st.v_init.get(t)
            if vi is not None:
                return self._dev(vi).reshape(st.shape).clone()
        return torch.zeros(st.shape, device=self.device)

    def iterate(self):
        """one iteration of the frame loop (styler_3p.py:301-386): a stylisation step per own key frame, the halo
        exchange of the updates, their temporal alignment, ``g_opt += `` -- returns the per-key-frame losses (device
        tensor, summed over ranks)"""
        st = self._st
        D, H, W_, _ = st.shape
        losses = torch.zeros(len(st.keys), device=self.device)
        upd = {}
        for j, t in enumerate(st.keys):
            if st.owner[t] == st.rank:
                slot = engine.optimizer_slot(getattr(self, "optimizer", "adam"), t, self.frames_per_opt)
                adam = st.opt_.get(slot)
                if adam is None:
                    adam = st.opt_[slot] = engine.make_optimizer(getattr(self, "optimizer", "adam"))
                st.work.copy_(st.g_opt[t])
                st.gs.bind(st.d[t], st.work.view(D, H, W_, 3) if self.target == "v" else st.work.view(D, H, W_), adam)
                losses[j] = st.gs.step(self._rot())
                dlt = torch.nan_to_num(st.gs.var.reshape(st.shape)) - st.g_opt[t]
                if self.target == "d":
                    dlt = dlt * st.d[t].reshape(st.shape)          # masking by original density (361-363)
                upd[t] = dlt
            # every rank draws the same view sequence whoever owns the frame (344-349)
            self._resample_views()
        if st.world > 1:
            parallel.all_reduce_sum_([losses], group=self.pg)
        if st.Wt is not None:
            got = parallel.exchange_frames(upd, st.need, st.owner, st.work, group=self.pg,
                                           need_by_rank=st.need_by_rank)
            for t in st.mine:
                st.g_opt[t] = st.g_opt[t] + self.aligned_update(t, got, st.u, st.Wt, st.keys)
        else:
            for t in st.mine:
                st.g_opt[t] = st.g_opt[t] + upd[t]
        st.hist.append(losses)
        return losses

    def finish(self):
        """frame interpolation (392-397) + final inference of every frame whose density this rank holds: all of them
        in an unsharded run (or a sharded one prepared with every frame), the rank's own frames after
        ``prepare(frames_on_device=...)`` -- ``result['frames']`` lists which entries of ``d`` / ``r`` / ``v`` these
        are"""
        st = self._st
        D, H, W_, _ = st.shape
        allv = parallel.exchange_frames(st.g_opt, set(st.keys), st.owner, st.work, group=self.pg,
                                        need_by_rank=[set(st.keys)] * st.world) if st.world > 1 else st.g_opt
        full = {t: allv[t] for t in st.keys}
        if self.interp > 1:
            w = np.linspace(0, 1, self.interp + 1)
            for t in range(0, st.F - 1, self.interp):
                for i in range(1, self.interp):
                    if t + self.interp < st.F:
                        full[t + i] = full[t] * float(1 - w[i]) + full[t + self.interp] * float(w[i])
        d_sty, r_sty, v_sty = [], [], []
        present = [t for t in range(st.F) if t in st.d]
        for t in present:
            var = full.get(t)
            if var is None:                                        # trailing frames past the last key frame
                var = self._initial(t)
            if self.target == "v":
                d_adv = ops.advect_fwd(st.d[t].unsqueeze(-1), var).squeeze(-1)
            else:
                d_adv = var.reshape(D, H, W_)
            d_out = ops.smooth3d_relu_fwd(d_adv.contiguous(), float(self.k))
            dimg = self.loss.d_img(d_out, self._identity)
            d_sty.append(torch.abs(d_out).unsqueeze(-1).cpu().numpy())   # abs(): drop the sign-bit mask of -0.0
            r_sty.append(dimg[0].cpu().numpy().astype(np.uint8))
            v_sty.append(var.cpu().numpy())
        hist = [[float(x) for x in l_.cpu()] for l_ in st.hist]
        return {"l": [[x for l_ in hist for x in l_]], "l_frames": hist, "d_intm": [],
                "d": np.array(d_sty), "r": np.array(r_sty), "v": v_sty if self.target == "v" else None,
                "opt": v_sty, "p": None, "c": None, "frames": present}

    def run(self, params):
        self.prepare(params)
        for _ in range(self.iter):
            self.iterate()
        return self.finish()


class argparse_ns(object):
    """plain attribute bag for the loop state"""
