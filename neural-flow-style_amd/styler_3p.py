"""3-D particle stylizer -- host-side mirror of the reference's ``styler_3p.Styler``
(styler_3p.py:14-439): same constructor / ``run(params) -> dict`` surface and the same
outer loop (octaves x iterations x frame batches x views, TF-Adam per ``opt_id``, update
masking, temporal Gaussian smoothing of the per-particle updates, frame interpolation,
final inference), driving the HIP kernels instead of a TF-1.15 graph.

Forward graph per frame (styler_3p.py:42-164):
  'd': r + clip(r_opt,-1,1) -> sum_k p2g_wavg(p, r_k, support / kernel_scale^k)
  'p': p + v              -> p2g(p) / rest_density        (+ pressure loss, 96-98)
  -> 3x3x3 smoothing conv -> max(.,0) = d_out -> [rotate] -> render -> max-normalise
  -> x255, grey->3ch = d_img -> VGG-19 -> Gram style loss (+TV).
The particle side runs through ``torch.autograd`` (custom Functions wrapping the splat
kernels); the render/VGG/loss side is the explicit ``engine.RenderStyleLoss`` chain.

Differences from the reference, all deliberate and documented:
  * ``views_mode='sum'`` (build extension): one Adam step on the gradient of the summed view
    losses; the reference-faithful default is ``'sequential'`` (one step per view-batch, iterates
    averaged, styler_3p.py:326-352).
  * the sqrt in the SPH kernel has zero gradient at 0 (analytic limit) instead of NaN-then-
    ``nan_to_num`` (styler_3p.py:337,340,360).
  * ``batch_size`` must be 1 (as every reference driver sets it): with batch_size>1 the reference
    couples the frames of a batch through the global max of the render (styler_3p.py:158).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import engine, ops, parallel
from . import transform as T
from .styler_base import StylerBase
from .util import temporal_weights


class _SmoothRelu(torch.autograd.Function):
    """conv3d SAME with [1,k,1]^3/(k+2)^3 then tf.maximum(d,0) (styler_3p.py:112-125)"""

    @staticmethod
    def forward(ctx, d, k):
        out = ops.smooth3d_relu_fwd(d.contiguous().reshape(d.shape[1:4]), k)
        ctx.k = k
        ctx.save_for_backward(out)
        ctx.shape = d.shape
        return out.reshape(d.shape)

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        gd = ops.smooth3d_relu_bwd(out, g.contiguous().reshape(out.shape), ctx.k)
        return gd.reshape(ctx.shape), None


class Styler(StylerBase):
    def __init__(self, self_dict):
        StylerBase.__init__(self, self_dict)
        assert self.batch_size == 1, "batch_size > 1 is not supported (see module docstring)"
        if self.rotate:
            self.rot_mat_, self.views = T.rot_mat(self.phi0, self.phi1, self.phi_unit, self.theta0, self.theta1,
                                                  self.theta_unit, sample_type=self.sample_type, rng=self.rng,
                                                  nv=self.n_views)
            if self.n_views is None:
                self.n_views = len(self.views)
            print("# vps:", self.n_views)
            assert self.n_views % self.v_batch == 0
        self.loss = self._make_loss(rotate=self.rotate)
        self._graph_loss = None                      # engine.GraphedLoss, decided at the first loss call
        self._identity = T.rot_to_device([np.identity(3)], self.device)
        # multi-GPU (set by the driver): ``pg`` = the process group; ``shard_by`` = 'views' (views=sum: the views of every
        # frame over the ranks, ONE all-reduce of the variable's gradient + loss per frame step) or 'frames' (SURVEY 8(e)
        # "Frames (config 4/5)": contiguous blocks of key frames per rank, full view batches per rank, and the only
        # exchange is the halo of per-frame updates the temporal filter reaches, point-to-point)
        self.pg = None
        self.shard_by = getattr(self, "shard_by", None) or "views"

    # ---- forward graph: variable -> d_out [1,D,H,W,1] (autograd) ---------------------------------
    def _field(self, p, r, var, res):
        """p [N,3], r [N,nk] device tensors; var = the optimised tensor ([N,3] or [N,nk]).
        Returns (positions, d_out, extra) with extra = the auxiliary loss terms on the field / variable
        (pressure loss, styler_base.py:226-230; density-preservation loss, styler_base.py:215-223) or None"""
        pressure = None
        extra = None
        p_ = p.unsqueeze(0)
        if "p" in self.target_field:
            p_ = p_ + var.unsqueeze(0)
        p_all = p_
        # a sequence keeps ONE particle order for all frames (frame 0's: the temporal filter needs particle i to be the
        # same particle everywhere), so by frame 60 of a flowing liquid the neighbours of that order have dispersed and
        # the splat would fall back to scattered global atomics; it therefore sees every frame through that frame's
        # own grid order (a gather of the positions in, the scatter of their gradient out -- the splat does not care
        # about the order of its particles)
        order = self._particle_order(p, p_)
        if order is not None:
            p_ = T.permute_particles(p_, order)
        if "d" in self.target_field:
            r_opt = torch.clamp(var.unsqueeze(0), -1, 1)                    # "necessary!" (styler_3p.py:74)
            r_ = r.unsqueeze(0) + r_opt
            if order is not None:
                r_ = T.permute_particles(r_, order)
            if getattr(self, "w_density", 0) > 0:
                # density preservation on the clipped offsets (self.d[i], styler_3p.py:75; styler_base.py:217-223)
                d_loss = r_opt[0].sum() ** 2
                d_pres = (-torch.log(r_opt[0].abs() + 1e-6)).sum()
                extra = (d_loss + d_pres * 1e3) * self.w_density
            d_ = None
            for k in range(self.num_kernels):
                support = self.support / self.kernel_scale ** k
                d_hat = T.p2g_wavg(p_, r_[..., k:k + 1], self.domain, res, self.radius, self.nsize, kernel="cubic",
                                   support=support, clip=self.clip, is_2d=False)
                d_ = d_hat if d_ is None else d_ + d_hat
        else:
            # d / rest_density (styler_3p.py:96): the particle mass is linear in rest_density (transform.py:1348-1352), so
            # the quotient is the splat with unit rest density -- no 32 MB scaling pass forward, none backward
            d_ = T.p2g(p_, self.domain, res, self.radius, 1.0, self.nsize, support=self.support,
                       clip=self.clip, is_2d=False)
            if self.w_pressure > 0:
                pressure = torch.where(d_ > 0, d_ - 1, torch.zeros_like(d_))
                extra = (pressure ** 2).mean() * self.w_pressure           # styler_base.py:228-230
        d_out = _SmoothRelu.apply(d_, float(self.k)) if self.k > 0 else _SmoothRelu.apply(d_, 0.0)
        return p_all[0], d_out, extra

    def _particle_order(self, p, p_now):
        """the permutation the splat sees frame ``p`` through, or None (= the caller's order).  Frames beyond the first
        get their own grid order at run start (``_orders``).  With the POSITIONS as the variable the particles drift away
        from the order they were put in -- 0.4 cells per step at lr 0.002 on a 200^3 grid; after 100 steps the boxes of
        consecutive particles no longer fit the splat's LDS accumulators and the forward splat runs at half speed
        (tools/chocolate_iter_profile.py: 213 us against 116 in grid order) -- so the order of a frame is recomputed from
        the CURRENT positions every ``reorder_every`` evaluations (default 10; 0 = never).  The splat does not care
        about the order of its particles; autograd scatters the gradient back through the gather.  The sort is STABLE
        (32-bit keys: the radix path either way), so two runs, or two ranks of a view-sharded run, that hold the same
        positions build the same layout and feed identically rounded sums to the blocks that fall back to float
        atomics."""
        orders = self.__dict__.setdefault("_orders", {})
        key = (p.data_ptr(), p.shape[0])
        every = int(getattr(self, "reorder_every", 10) or 0)
        if "p" not in self.target_field or every <= 0 or not getattr(self, "sort_particles", True) or p.shape[0] < 2:
            return orders.get(key)
        ages = self.__dict__.setdefault("_order_age", {})
        age = ages.get(key, 0) + 1
        if age >= every:
            orders[key] = T.grid_order(p_now[0].detach(), self.resolution, stable=True)
            age = 0
        ages[key] = age
        return orders.get(key)

    def _value_and_grad(self, p, r, var, res, rot, view_shard=True):
        """loss (per view, device) and d loss / d var for one frame.  ``view_shard``: the views are sharded over the
        ranks of ``self.pg`` (then only rank 0 adds the view-independent terms)"""
        v = var.detach().clone().requires_grad_(True)
        _, d_out, extra = self._field(p, r, v, res)
        d3 = d_out.detach().reshape(d_out.shape[1:4]).contiguous()
        nviews = int(rot.shape[0]) if (self.rotate and rot is not None) else 1
        if self._graph_loss is None:
            # hipGraph replay of the loss chain where the host cannot keep up with it: tried (and timed against eager
            # submission) with one or two views per call; NFS_GRAPH=0 / 1 forces it off / on.  The GraphedLoss object is
            # made once; whether THIS call goes through it depends on this call's view count (it keys its capture on
            # the shapes and starts over when they change)
            env = os.environ.get("NFS_GRAPH")
            self._graph_loss = (engine.GraphedLoss(self.loss, force=True) if env == "1" else
                                engine.GraphedLoss(self.loss) if env is None else False)
        if self._graph_loss and (self._graph_loss.force or nviews <= 2):
            losses, g_d = self._graph_loss(d3, rot)
            losses = losses.clone()
        elif self.loss.writes_gradient(nviews):
            g_d = torch.empty_like(d3)                       # (written, not accumulated into: no zero fill)
            losses = self.loss.loss_and_grad(d3, rot, g_d, overwrite=True)
        else:
            g_d = torch.zeros_like(d3)
            losses = self.loss.loss_and_grad(d3, rot, g_d)
        # view-independent terms (pressure / density preservation) belong to the iteration, not to a view: with the
        # views sharded over ranks (views=sum) only rank 0 adds them, so that the all-reduced loss and gradient
        # contain them ONCE (every rank would otherwise contribute a copy: world x the single-rank weight)
        if extra is not None and view_shard and self._rank_world()[0] != 0:
            extra = None
        if extra is not None:
            losses = losses + extra.detach() / losses.numel()
        heads, grads = [d_out], [g_d.reshape(d_out.shape)]
        if extra is not None:
            heads.append(extra); grads.append(torch.ones_like(extra))
        torch.autograd.backward(heads, grads)
        return losses, v.grad

    def _rot(self, lo, hi):
        return T.rot_to_device(self.rot_mat_[lo:hi], self.device)

    def _resample_views(self):
        if "uniform" not in self.sample_type:
            self.rot_mat_, self.views = T.rot_mat(self.phi0, self.phi1, self.phi_unit, self.theta0, self.theta1,
                                                  self.theta_unit, sample_type=self.sample_type, rng=self.rng,
                                                  nv=self.n_views)

    def _dev(self, a):
        return torch.as_tensor(np.asarray(a, np.float32)).to(self.device).contiguous()

    # ---- debug helper: rendered images of the un-stylised input ------------------------------------
    def render_test(self, params):
        imgs = []
        for t in range(self.num_frames):
            p = self._dev(params["p"][t])
            n = p.shape[0]
            r = self._dev(params["r"][t]) if "d" in self.target_field else None
            var = torch.zeros(n, 3 if "p" in self.target_field else self.num_kernels, device=self.device)
            with torch.no_grad():
                _, d_out, _ = self._field(p, r, var, list(self.resolution))
                dimg = self.loss.d_img(d_out.reshape(d_out.shape[1:4]).contiguous(), self._identity)
            imgs.append(dimg[0].cpu().numpy().astype(np.uint8))
        return imgs

    # ---- the optimisation loop (styler_3p.py:229-439) ----------------------------------------------
    def run(self, params):
        if abs(self.lr_scale - 1) > 1e-7 and not isinstance(self.lr, list):
            self.lr = [self.lr / self.lr_scale ** i for i in range(self.octave_n)]

        oct_size = []
        dhw = np.array(self.resolution)
        for _ in range(self.octave_n):
            oct_size.append(dhw)
            dhw = (dhw // self.octave_scale).astype(int)
        oct_size.reverse()
        print("input size for each octave", oct_size)

        p = [self._dev(x) for x in params["p"]]
        r = [self._dev(x) for x in params["r"]] if "d" in self.target_field else [None] * self.num_frames
        # Particles in grid-cell order (one permutation for all frames, from the first frame's positions: the temporal
        # filter and the frame interpolation need particle i to be the same particle in every frame).  A block of
        # consecutive particles then touches a small box of cells and the splat accumulates it in LDS
        # (5e5 particles on 200^3: 0.69 ms unordered, 0.37 ordered with global atomics, 0.16 in LDS); the outputs are
        # returned in the caller's order.
        inv = None
        if getattr(self, "sort_particles", True) and len(set(int(x.shape[0]) for x in p)) == 1 and p[0].shape[0] > 1:
            perm = T.grid_order(p[0], self.resolution)
            inv = torch.empty_like(perm)
            inv[perm] = torch.arange(perm.numel(), device=self.device)
            p = [x[perm].contiguous() for x in p]
            r = [x[perm].contiguous() if x is not None else None for x in r]
        self._orders, self._order_age = {}, {}
        if getattr(self, "sort_particles", True) and self.num_frames > 1:
            for x in p[1:]:                                  # (frame 0 is in its own order already)
                if x.shape[0] > 1:
                    self._orders[(x.data_ptr(), x.shape[0])] = T.grid_order(x, self.resolution)
        nvar = 3 if "p" in self.target_field else self.num_kernels
        g_opt = [torch.zeros(p[i].shape[0], nvar, device=self.device) for i in range(self.num_frames)]
        mode = getattr(self, "views_mode", "sequential")
        if getattr(self, "optimizer", "adam") == "lbfgs" and self.rotate and mode == "sequential" \
                and self.n_views > self.v_batch:
            raise ValueError("optimizer=lbfgs with views_mode='sequential': every step sees a different view batch, i.e. "
                             "a different objective -- gradient differences across them are not curvature pairs.  Use "
                             "views_mode='sum' (one objective per iteration) or optimizer=adam")

        # key frames (304-309) and who stylises them.  Frame sharding follows SURVEY 8(e): contiguous blocks of optimiser
        # groups per rank (one Adam state never straddles ranks), every rank keeps full view batches for its frames.
        keys = list(range(0, self.num_frames, self.batch_size * self.interp))
        rank, world = self._rank_world()
        by_frames = self.pg is not None and world > 1 and self.shard_by == "frames"
        if by_frames:
            plan = parallel.plan_frames(self.num_frames, self.batch_size * self.interp, self.frames_per_opt, world)
            owner = {t: rk for rk, ts in enumerate(plan) for t in ts}
        else:
            owner = {t: rank for t in keys}
        mine = [t for t in keys if owner[t] == rank]
        # the temporal Gaussian over the per-frame updates (382-383: scipy gaussian_filter along the frame axis, reflect,
        # truncate 4 sigma) as its matrix (util.temporal_weights == util.denoise, pinned to the reference's own function
        # by tests/golden/util_reference.npz): evaluated on the device, and -- frames sharded -- only the frames inside a
        # rank's filter window travel (point-to-point halo, parallel.exchange_frames)
        Wt = temporal_weights(len(keys), self.window_sigma) if (self.window_sigma > 0 and len(keys) > 1) else None
        if Wt is not None:
            assert len(set(int(x.shape[0]) for x in p)) == 1, "temporal smoothing needs the same particles in every frame"
        need_by_rank = []
        for rk in range(world):
            nd = set(t for t in keys if owner[t] == rk) if by_frames else set(keys)
            if Wt is not None:
                for t in list(nd):
                    nd |= set(keys[jj] for jj in np.nonzero(Wt[keys.index(t)])[0])
            need_by_rank.append(nd)
        like = torch.zeros(p[0].shape[0], nvar, device=self.device)

        def sync_all(g):
            """every rank receives every key frame's variable (octave ends, final inference: a few MB per frame)"""
            if not by_frames:
                return g
            got = parallel.exchange_frames({t: g[t] for t in mine}, set(keys), owner, like, group=self.pg,
                                           need_by_rank=[set(keys)] * world)
            for t in keys:
                g[t] = got[t]
            return g

        loss_history, d_intm, opt_ = [], [], {}
        for octave in range(self.octave_n):
            loss_history_o, d_intm_o = [], []
            res = [int(v) for v in oct_size[octave]]
            if self.style_img is not None:
                self.loss.set_style_image(self._style_feature(self.style_img, res[1:]))
                if getattr(self, "w_hist", 0) > 0:                # styler_3p.py:288-293
                    self.loss.set_hist_image(self._hist_feature(self.style_img, res[1:]))
            if self.content_img is not None:                     # styler_3p.py:277-279
                self.loss.set_content_image(self._content_feature(self.content_img, res[1:]), top_k=self._content_top_k())
            lr = self.lr[octave] if isinstance(self.lr, list) else self.lr

            for step in range(self.iter):
                g_tmp = {}
                l_step = torch.zeros(len(keys), device=self.device)
                for j, t in enumerate(keys):
                    if owner[t] != rank:
                        # every rank draws the same view sequence whoever owns the frame (344-349)
                        if self.rotate:
                            self._resample_views()
                        continue
                    var = g_opt[t].clone()                       # variable re-assigned from g_opt (312)
                    opt_id = engine.optimizer_slot(getattr(self, "optimizer", "adam"), t, self.frames_per_opt)
                    if opt_id not in opt_:
                        opt_[opt_id] = engine.make_optimizer(getattr(self, "optimizer", "adam"))
                    adam = opt_[opt_id]

                    if self.rotate and mode == "sequential":
                        acc, l_ = None, []
                        for i in range(0, self.n_views, self.v_batch):
                            losses, g = self._value_and_grad(p[t], r[t], var, res, self._rot(i, i + self.v_batch),
                                                             view_shard=False)     # (the sequential mode never shards views)
                            adam.step(var, g.contiguous(), lr)
                            l_.append(losses.sum())
                            cur = torch.nan_to_num(var)
                            acc = cur.clone() if acc is None else acc + cur
                        l_step[j] = torch.stack(l_).mean()
                        self._resample_views()
                        new = acc / (self.n_views / self.v_batch)
                    else:
                        if self.rotate:
                            nvw = len(self.rot_mat_)
                            rot = self._rot(0, nvw)
                            if self.pg is not None and not by_frames:
                                rot = rot[rank::world].contiguous()
                        else:
                            rot = self._identity
                        losses, g = self._value_and_grad(p[t], r[t], var, res, rot,
                                                         view_shard=self.pg is not None and not by_frames)
                        total = losses.sum()
                        if self.pg is not None and not by_frames:
                            g = g.contiguous()
                            parallel.all_reduce_sum_([g, total], group=self.pg)
                        adam.step(var, g.contiguous(), lr)
                        l_step[j] = total
                        if self.rotate:
                            self._resample_views()
                        new = torch.nan_to_num(var)

                    upd = new - g_opt[t]
                    if "d" in self.target_field:
                        upd = upd * r[t][..., 0:1]                  # masking by original density (361-363)
                    g_tmp[t] = upd

                    if step == self.iter - 1 and octave < self.octave_n - 1 and not by_frames:
                        with torch.no_grad():
                            _, d_out, _ = self._field(p[t], r[t], var, res)
                            dimg = self.loss_d_img(d_out)
                        d_intm_o.append(dimg.cpu().numpy().astype(np.uint8))

                if by_frames:
                    parallel.all_reduce_sum_([l_step], group=self.pg)
                loss_history_o.extend(float(x) for x in l_step.cpu())
                if Wt is not None:
                    got = parallel.exchange_frames(g_tmp, need_by_rank[rank], owner, like, group=self.pg,
                                                   need_by_rank=need_by_rank) if by_frames else g_tmp
                    for t in mine:
                        jrow = Wt[keys.index(t)]
                        f = None
                        for jj in np.nonzero(jrow)[0]:
                            term = got[keys[jj]] * float(jrow[jj])
                            f = term if f is None else f.add_(term)
                        g_opt[t] = g_opt[t] + f
                else:
                    for t in mine:
                        g_opt[t] = g_opt[t] + g_tmp[t]

            loss_history.append(loss_history_o)
            if by_frames:
                g_opt = sync_all(g_opt)
                if octave < self.octave_n - 1:                   # (every rank renders every key frame: same d_intm everywhere)
                    for t in keys:
                        with torch.no_grad():
                            _, d_out, _ = self._field(p[t], r[t], g_opt[t], res)
                            dimg = self.loss_d_img(d_out)
                        d_intm_o.append(dimg.cpu().numpy().astype(np.uint8))
            if octave < self.octave_n - 1:
                d_intm.append(np.concatenate(d_intm_o, axis=0))

        if self.interp > 1:
            w = np.linspace(0, 1, self.interp + 1)
            for t in range(0, self.num_frames - 1, self.interp):
                for i in range(1, self.interp):
                    if t + self.interp < self.num_frames:
                        g_opt[t + i] = g_opt[t] * float(1 - w[i]) + g_opt[t + self.interp] * float(w[i])

        result = {"l": loss_history, "d_intm": d_intm, "v": None, "c": None}
        res = [int(v) for v in oct_size[-1]]
        p_sty, v_sty, d_sty, r_sty = [], [], [], []
        for t in range(self.num_frames):
            with torch.no_grad():
                p_out, d_out, _ = self._field(p[t], r[t], g_opt[t], res)
                dimg = self.loss_d_img(d_out)
            p_sty.append((p_out if inv is None else p_out[inv]).cpu().numpy())
            if "p" in self.target_field:
                v_sty.append((g_opt[t] if inv is None else g_opt[t][inv]).cpu().numpy())
            d_sty.append(torch.abs(d_out[0]).cpu().numpy())      # abs(): drop the sign-bit mask of -0.0
            r_sty.append(dimg[0].cpu().numpy().astype(np.uint8))
        result["p"] = p_sty
        if "p" in self.target_field:
            result["v"] = v_sty
        result["d"] = np.array(d_sty)
        result["r"] = np.array(r_sty)
        result["opt"] = [(g if inv is None else g[inv]).cpu().numpy() for g in g_opt]   # build extension: the variables
        self._orders, self._order_age = {}, {}   # keyed by device address: the frame tensors die with this call
        return result

    def loss_d_img(self, d_out):
        """d_img with the identity rotation (styler_3p.py:366-369, 416-420)"""
        d3 = d_out.reshape(d_out.shape[1:4]).contiguous()
        return self.loss.d_img(d3, self._identity)

    def _rank_world(self):
        return (0, 1) if self.pg is None else parallel.rank_world(self.pg)
