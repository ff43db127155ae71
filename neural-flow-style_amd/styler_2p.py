"""2-D colour stylizer -- host-side mirror of the reference's ``styler_2p.Styler``
(styler_2p.py:14-315; BASELINE config 1, dambreak2d): per-particle colour ``c`` [N,3] is the
Adam variable, splatted to an image by the SPH colour splat ``p2g(p, pc=c, pd=r)``.

Forward graph (styler_2p.py:42-102):
  d_gray = clip(p2g(p) / rest_density, 0, 1)            (mask; independent of the variable)
  d      = clip(p2g(p, pc=clip(c,0,1), pd=r), 0, 1)     [1,H,W,3]
  d_img  = d * 255 -> VGG -> Gram style loss (optionally masked by d_gray) + TV
  d_out  = d * d_gray  (the returned image)
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import engine
from . import ops
from . import transform as T
from . import vgg as vggmod
from .styler_base import StylerBase
from .util import denoise


class Styler(StylerBase):
    def __init__(self, self_dict):
        StylerBase.__init__(self, self_dict)
        self.batch_size = max(int(self.batch_size), 1)
        if self.batch_size > 1 and getattr(self, "w_hist", 0):
            raise NotImplementedError("batch_size > 1 with a histogram term: the match runs over the whole batch "
                                      "tensor (styler_base.py:203-207)")
        w_layers = list(self.w_style_layer)
        if len(w_layers) == 1 and len(self.style_layer) > 1:
            w_layers = w_layers * len(self.style_layer)
        self.loss = engine.ImageStyleLoss(self.net, self.style_layer, w_layers, self.w_style, w_tv=self.w_tv,
                                          resize_scale=self.resize_scale, style_mask=self.style_mask,
                                          style_mask_on_ref=self.style_mask_on_ref,
                                          w_content=getattr(self, "w_content", 0),
                                          content_layer=getattr(self, "content_layer", None),
                                          content_channel=getattr(self, "content_channel", 0),
                                          w_content_amp=getattr(self, "w_content_amp", 100),
                                          w_hist=getattr(self, "w_hist", 0), hist_layer=getattr(self, "hist_layer", ()),
                                          w_hist_layer=getattr(self, "w_hist_layer", ()))

    def _dev(self, a):
        return torch.as_tensor(np.asarray(a, np.float32)).to(self.device).contiguous()

    def _order(self, p):
        """the frame's own grid order (computed once per frame and run): the splat accumulates a block of consecutive
        particles in LDS when their cells sit in a small box, whatever order the caller's particles come in -- a
        gather of positions / colours in, the scatter of the colour gradient out"""
        if not getattr(self, "sort_particles", True) or p.shape[0] < 2:
            return None
        cache = self.__dict__.setdefault("_orders", {})
        key = (p.data_ptr(), p.shape[0])
        if key not in cache:
            cache[key] = T.grid_order(p, self.resolution)
        assert cache[key].numel() == p.shape[0]
        return cache[key]

    def _density(self, p, res):
        o = self._order(p)
        pp = p if o is None else p[o]
        d = T.p2g(pp.unsqueeze(0), self.domain, res, self.radius, self.rest_density, self.nsize,
                  support=self.support, clip=self.clip)
        return torch.clamp(d / self.rest_density, 0, 1)                      # [1,H,W,1]

    def _colour(self, p, r, var, res):
        c_ = torch.clamp(var.unsqueeze(0), 0, 1)
        o = self._order(p)
        # (the colours through the permutation and back: the adjoint of a gather through a permutation is a plain
        # scatter, transform._Permute)
        pp, cc, rr = (p, c_, r) if o is None else (p[o], T.permute_particles(c_, o), r[o])
        d = T.p2g(pp.unsqueeze(0), self.domain, res, self.radius, self.rest_density, self.nsize,
                  support=self.support, clip=self.clip, pc=cc, pd=rr.unsqueeze(0))
        return torch.clamp(d, 0, 1), c_[0]                                   # [1,H,W,3]

    def _colour_ordered(self, frame, var, res):
        """``_colour`` on a frame already brought into its grid order (``frame`` = (order, positions [1,N,2], densities
        [1,N,1], mask): what run() prepares once per octave); only the colours go through the permutation per step"""
        o, pp, rr, _ = frame
        c_ = torch.clamp(var.unsqueeze(0), 0, 1)
        cc = c_ if o is None else T.permute_particles(c_, o)
        d = T.p2g(pp, self.domain, res, self.radius, self.rest_density, self.nsize, support=self.support, clip=self.clip,
                  pc=cc, pd=rr)
        return torch.clamp(d, 0, 1)                                          # [1,H,W,3]

    def render_test(self, params):
        res = list(self.resolution)
        out = []
        for t in range(self.num_frames):
            p, r = self._dev(params["p"][t]), self._dev(params["r"][t])
            with torch.no_grad():
                d, _ = self._colour(p, r, torch.full((p.shape[0], 3), 0.5, device=self.device), res)
                out.append((d * self._density(p, res))[0].cpu().numpy())
        return out

    def run(self, params):
        oct_size = []
        hw = np.array(self.resolution)
        for _ in range(self.octave_n):
            oct_size.append(hw)
            hw = (hw // self.octave_scale).astype(int)
        oct_size.reverse()
        print("input size for each octave", oct_size)

        p = [self._dev(x) for x in params["p"]]
        r = [self._dev(x) for x in params["r"]]
        self._orders = {}
        self._graph_loss = None
        n = p[0].shape[0]
        # colour init: noise around the VGG mean / 255 (styler_2p.py:189-192)
        c_opt = self.rng.uniform(-5, 5, [self.num_frames, n, 3]).astype(np.float32)
        c_opt += np.array([vggmod._R_MEAN, vggmod._G_MEAN, vggmod._B_MEAN], np.float32)
        c_opt /= 255
        g_opt = [self._dev(c_opt[i]) for i in range(self.num_frames)]

        loss_history, d_intm, opt_ = [], [], {}
        style_per_octave = []
        for octave in range(self.octave_n):
            loss_history_o, d_intm_o = [], []
            res = [int(v) for v in oct_size[octave]]
            if self.style_img is not None:
                style_o = self._style_feature(self.style_img, res)
                style_per_octave.append(np.asarray(style_o, np.float32))
                self.loss.set_style_image(style_o)
                if getattr(self, "w_hist", 0) > 0:               # styler_2p.py:220-225
                    self.loss.set_hist_image(self._hist_feature(self.style_img, res))
            if self.content_img is not None:                     # styler_2p.py:209-211
                self.loss.set_content_image(self._content_feature(self.content_img, res), top_k=self._content_top_k())
            lr = self.lr[octave] if isinstance(self.lr, list) else self.lr
            # what an iteration does not change is formed once per octave: every frame in its grid order (a gather of
            # positions and densities per step otherwise) and its density mask (a splat + two scalings per step)
            frames = []
            for t in range(self.num_frames):
                o = self._order(p[t])
                pp, rr = (p[t], r[t]) if o is None else (p[t][o], r[t][o])
                with torch.no_grad():
                    frames.append((o, pp.unsqueeze(0), rr.unsqueeze(0), self._density(p[t], res)))
            nb = self.num_frames // max(self.batch_size, 1)
            # the per-step losses stay on the device until the octave is done: reading one back per step is a host
            # synchronisation per step, and the whole iteration is ~0.3 ms of kernels
            loss_dev = torch.zeros(self.iter * max(nb, 1), dtype=torch.float32, device=self.device)
            # The chain of one frame is clip -> (permutation) -> splat -> clip -> loss net: five small launches either side
            # of the loss chain.  With torch autograd each of them is a node (the two clips' adjoints alone are eight
            # elementwise kernels) and the iteration is host-bound; written out by hand on the C-ABI operators
            # (nfs_colour_clamp_gather / nfs_p2g_fwd / nfs_clamp01_bwd / nfs_p2g_bwd / nfs_colour_clamp_scatter_bwd /
            # nfs_iterate_update) it is the same arithmetic in a third of the launches (NFS_2P_AUTOGRAD=1: the autograd form)
            explicit = os.environ.get("NFS_2P_AUTOGRAD", "0") != "1"
            cfg = ops.make_splat_cfg(2, list(res), list(self.domain), self.radius, self.support, self.rest_density,
                                     self.nsize, self.clip, 1)
            filt = self.window_sigma > 0 and self.num_frames > 1
            vars_dev = [g.clone() for g in g_opt] if explicit else None      # the variables ApplyAdam updates in place
            for step in range(self.iter):
                g_tmp = [None] * self.num_frames
                B = self.batch_size
                assert self.num_frames % B == 0, "num_frames must be a multiple of batch_size (styler_2p.py:239-244)"
                for t in range(0, self.num_frames, B):
                    # B frames share one sess.run (styler_2p.py:42-98: batch_size towers): one image batch through the
                    # loss network -- style summed over the images, TV averaged (styler_base.py:181, 212) -- and ONE
                    # optimiser step on the B colour variables (Adam slots per batch position, one step count)
                    opt_id = engine.optimizer_slot(getattr(self, "optimizer", "adam"), t, self.frames_per_opt)
                    if opt_id not in opt_:
                        opt_[opt_id] = engine.make_optimizer(getattr(self, "optimizer", "adam"))
                    if explicit:
                        ccs = [ops.colour_clamp_gather(vars_dev[t + i], frames[t + i][0]) for i in range(B)]
                        raws = [ops.p2g_fwd(frames[t + i][1][0], cfg, attr=ccs[i], pd=frames[t + i][2].reshape(-1))
                                for i in range(B)]
                        d_raw = raws[0].unsqueeze(0) if B == 1 else torch.stack(raws)
                        d = torch.clamp(d_raw, 0, 1)
                    else:
                        vars_ = [g_opt[t + i].clone().requires_grad_(True) for i in range(B)]
                        imgs = [self._colour_ordered(frames[t + i], vars_[i], res) for i in range(B)]
                        d = imgs[0] if B == 1 else torch.cat(imgs, 0)
                    d_gray = frames[t][3] if B == 1 else torch.cat([frames[t + i][3] for i in range(B)], 0)
                    if self._graph_loss is None:
                        # the loss chain of one colour image is ~40 small launches: hipGraph replay where a measured
                        # trial finds the host cannot keep up with it (engine.GraphedLoss; NFS_GRAPH=0 / 1 forces)
                        env = os.environ.get("NFS_GRAPH")
                        self._graph_loss = (engine.GraphedLoss(self.loss, force=True) if env == "1" else
                                            engine.GraphedLoss(self.loss) if env is None else False)
                    if self._graph_loss:
                        losses, g_d = self._graph_loss(d.detach().contiguous(), d_gray)
                    else:
                        losses, g_d = self.loss.loss_and_grad(d.detach().contiguous(), d_gray)
                    k = step * nb + t // B
                    torch.sum(losses, dim=0, keepdim=True, out=loss_dev[k:k + 1])    # (before the next call overwrites them)
                    if explicit:
                        g_raw = ops.clamp01_bwd(g_d.contiguous(), d_raw)
                        gxs = []
                        for i in range(B):
                            _, g_cc, _ = ops.p2g_bwd(frames[t + i][1][0], cfg, g_raw[i], attr=ccs[i],
                                                     pd=frames[t + i][2].reshape(-1), need_p=False, need_attr=True)
                            gxs.append(ops.colour_clamp_scatter_bwd(g_cc, vars_dev[t + i], frames[t + i][0]))
                        if B == 1:
                            x, gx = vars_dev[t].unsqueeze(0), gxs[0].unsqueeze(0)
                        else:
                            x, gx = torch.stack(vars_dev[t:t + B]), torch.stack(gxs)
                    else:
                        d.backward(g_d)
                        if B == 1:                                                   # (views: nothing to stack)
                            x, gx = vars_[0].detach().unsqueeze(0), vars_[0].grad.unsqueeze(0)
                        else:
                            x = torch.stack([v.detach() for v in vars_])            # [B,N,3]
                            gx = torch.stack([v.grad for v in vars_])
                    opt_[opt_id].step(x, gx.contiguous(), lr)
                    if step == self.iter - 1 and octave < self.octave_n - 1:
                        # the octave's intermediate image is d_out of the sess.run that applied the step
                        # (styler_2p.py:264-266): rendered from the RAW Adam-updated variable, before the iterate
                        # update below sanitises it (x is a view of the variable on the explicit path)
                        with torch.no_grad():
                            dd = torch.cat([self._colour(p[t + i], r[t + i], x[i], res)[0] for i in range(B)], 0)
                            d_intm_o.append(((dd * d_gray) * 255).cpu().numpy().astype(np.uint8))
                    for i in range(B):
                        if explicit and not filt and B == 1:
                            # g_opt += nan_to_num(x) - g_opt, and the variable restarts from it: one launch
                            ops.iterate_update(vars_dev[t + i], g_opt[t + i])
                        else:
                            g_tmp[t + i] = torch.nan_to_num(x[i]) - g_opt[t + i]
                if filt:
                    stack = denoise(np.stack([g.cpu().numpy() for g in g_tmp]), sigma=(self.window_sigma, 0, 0))
                    g_tmp = [self._dev(s) for s in stack]
                for t in range(self.num_frames):
                    if g_tmp[t] is not None:
                        g_opt[t] = g_opt[t] + g_tmp[t]
                        if explicit:
                            vars_dev[t].copy_(g_opt[t])
            loss_history_o = [float(v) for v in loss_dev.cpu().numpy()]
            loss_history.append(loss_history_o)
            if octave < self.octave_n - 1:
                d_intm.append(np.concatenate(d_intm_o, axis=0))

        result = {"l": loss_history, "d_intm": d_intm}
        res = [int(v) for v in oct_size[-1]]
        c_sty, d_sty = [], []
        for t in range(self.num_frames):
            with torch.no_grad():
                d, c_ = self._colour(p[t], r[t], g_opt[t], res)
                d_gray = self._density(p[t], res)
                c_sty.append((c_ * torch.clamp(r[t] / self.rest_density, 0, 1)).cpu().numpy())
                d_sty.append(((d * d_gray)[0] * 255).cpu().numpy().astype(np.uint8))
        result["c"] = c_sty
        result["d"] = np.array(d_sty)
        result["opt"] = [g.cpu().numpy() for g in g_opt]
        # extras of this build (not in the reference's dict): what the run started from, for parity tests
        result["c_init"] = c_opt
        result["style_per_octave"] = style_per_octave
        self._orders = {}        # keyed by device address: the frame tensors die with this call, the addresses get re-used
        return result
