"""The stylisation step: density volume -> (rotate+)render -> loss net -> Gram style
loss, and the whole adjoint chain back to the volume, on the HIP kernels.

This is the body of one ``sess.run(train_op)`` of the reference (styler_3p.py:147-164 +
styler_base.py:33-57,127-231) with views batched: all V local views go through the
renderer and VGG as one batch, the field gradient accumulates over views in one buffer.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import ops, parallel


def _channels(acts, name, F):
    """logical channel count of a loss-network tensor: Inception module outputs come in rows padded to a multiple of 64
    floats (``acts.channels``); VGG activations are what their shape says"""
    ch = getattr(acts, "channels", None)
    return int(ch[name]) if ch is not None and name in ch else int(F.shape[-1])


def _post_relu(name):
    """True for tensors that are ReLU outputs (or pools / concatenations of them): a gradient injected there may carry
    the ReLU mask.  The Inception graph's '*_pre_relu' tensors are not."""
    return not (name.endswith("_pre_relu") or name.endswith("_pre_activation"))


def _logical(acts, name):
    """(contiguous tensor of the logical channels, padded?)"""
    F = acts[name]
    c = _channels(acts, name, F)
    return (F, False) if c == F.shape[-1] else (F[..., :c].contiguous(), True)


def _add_logical(sg, name, F, g_log):
    """add a gradient over the logical channels into the (padded) entry of the adjoint chain"""
    g = sg.get(name)
    if g is None:
        g = sg[name] = torch.zeros_like(F)
    g[..., :g_log.shape[-1]].add_(g_log)


class RenderStyleLoss(object):
    """loss(d) = w_style * sum_v sum_l w_l * || G_l(vgg(render(rotate(d, R_v)))) - G_l^style ||^2
    (+ w_tv * TV).  ``loss_and_grad`` returns the per-view losses and ADDS dL/dd to ``g_d``."""

    def __init__(self, net, style_layer, w_style_layer, w_style=1.0, transmit=0.01, render_liquid=False,
                 resize_scale=1.0, rotate=True, w_tv=0.0, v_batch=1, w_content=0.0, content_layer=None,
                 content_channel=0, w_content_amp=100.0, w_hist=0.0, hist_layer=(), w_hist_layer=(), ray_mode=""):
        self.net = net
        # styler_base.py:152: the style term exists only with w_style != 0 (the semantic-transfer runs of run.bat:14-20
        # have w_style 0, the flag default, and no style image)
        self.layers = list(style_layer) if float(w_style) else []
        self.w_layers = [float(w) for w in w_style_layer][:len(self.layers)] if self.layers else []
        assert len(self.layers) == len(self.w_layers)
        self.w_style = float(w_style)
        self.tau = float(transmit)
        # ray integral: the reference's transmittance (smoke, max-normalised) / liquid forms, or -- north_star's "per-ray
        # max / mean" -- reduce_max (the line the reference keeps commented out, styler_3p.py:149) / reduce_mean along the
        # ray; like the liquid form these are not max-normalised
        if ray_mode not in ("", None, "transmit", "liquid", "max", "mean"):
            raise ValueError("ray_mode %r: '', 'transmit', 'liquid', 'max' or 'mean'" % (ray_mode,))
        self.mode = {"max": 2, "mean": 3, "liquid": 1, "transmit": 0}.get(ray_mode or "", 1 if render_liquid else 0)
        self.resize_scale = float(resize_scale)
        self.rotate = bool(rotate)
        self.w_tv = float(w_tv)
        self.v_batch = int(v_batch)
        self.two_pass_adjoint = True   # False: single fused adjoint with global atomics (less memory)
        # optional: view groups that run concurrently through VGG on separate HIP streams (>= 4 local views).
        # With the direct conv kernels two groups overlapped each other's launch tails (7.32 -> 7.01 ms/step);
        # with the Winograd path one batch of all views on one stream is fastest (8 views of 200^2: 1 stream
        # 4.73 ms/step, 2 streams 4.79, 4 streams 6.38), so that is the default.
        # all style layers' Gram work as three launches after the forward pass (default) / per layer (side stream)
        self.gram_grouped = os.environ.get("NFS_GRAM_GROUP", "1") != "0"
        self.gram_side_stream = os.environ.get("NFS_GRAM_STREAM", "1") != "0"
        self._side = None
        self.vgg_streams = int(os.environ.get("NFS_VGG_STREAMS", "1"))
        self.view_groups = int(os.environ.get("NFS_VIEW_GROUPS", "1"))
        self._streams = []
        order = [s[0] for s in net.seq]
        # content term on a layer of the same network (styler_base.py:135-150): channel maximisation, -mean, or the
        # distance to a content image's features (set_content_image)
        self.w_content = float(w_content) if content_layer else 0.0
        self.content_layer = content_layer if self.w_content else None
        self.content_channel = int(content_channel or 0)
        self.w_content_amp = float(w_content_amp)
        self.content_feature = None
        if self.content_layer is not None:
            kinds = dict((s[0], s[1]) for s in net.seq)
            if kinds.get(self.content_layer) != "conv":
                raise KeyError("content_layer %r is not a conv layer of the loss network" % (self.content_layer,))
        # histogram term (styler_base.py:187-209): layers of the loss network and / or 'input' (= d_img)
        self.w_hist = float(w_hist)
        self.hist_layers = list(hist_layer) if self.w_hist else []
        self.w_hist_layers = [float(w) for w in w_hist_layer] if self.w_hist else []
        if len(self.w_hist_layers) == 1 and len(self.hist_layers) > 1:
            self.w_hist_layers = self.w_hist_layers * len(self.hist_layers)
        if len(self.w_hist_layers) != len(self.hist_layers):
            raise ValueError("w_hist_layer has %d entries for %d hist layers (one weight, or one per layer)"
                             % (len(self.w_hist_layers), len(self.hist_layers)))
        for name in self.hist_layers:
            if "input" not in name and name not in order:
                raise KeyError("hist_layer %r is not a layer of the loss network" % (name,))
        self.hist_targets = None
        vgg_hist = [n for n in self.hist_layers if "input" not in n]
        wanted = self.layers + vgg_hist + ([self.content_layer] if self.content_layer else [])
        if not wanted:
            raise ValueError("no loss term reaches the loss network: w_style 0 (or no style layer) and no content / "
                             "histogram layer")
        for n in wanted:
            if n not in order:
                raise KeyError("%r is not a tensor of the loss network" % (n,))
        self.top = max(wanted, key=order.index)
        self.style_grams = None

    @property
    def liquid(self):
        """True for every ray integral that is not max-normalised (all but the transmittance form)"""
        return self.mode != 0

    @liquid.setter
    def liquid(self, on):
        self.mode = (1 if on else 0) if self.mode < 2 else self.mode

    def set_hist_image(self, style_img):
        """template features of the histogram term: style_img float32 [h,w,3] in 0..255 at the loss-net input size
        (styler_base.py:280-309: the style image fed at d_img, each hist layer fetched)"""
        if not self.hist_layers:
            self.hist_targets = None
            return None
        dev = self.net.device
        s = torch.as_tensor(np.asarray(style_img, np.float32)).to(dev)
        mean = torch.tensor([0.485 * 255, 0.456 * 255, 0.406 * 255], dtype=torch.float32, device=dev)
        vgg_hist = [n for n in self.hist_layers if "input" not in n]
        acts = {}
        if vgg_hist:
            order = [q[0] for q in self.net.seq]
            acts = self.net.forward((s - mean).unsqueeze(0).contiguous(), max(vgg_hist, key=order.index))
        self.hist_targets = {n: (s.unsqueeze(0).contiguous() if "input" in n else _logical(acts, n)[0].clone())
                             for n in self.hist_layers}
        return self.hist_targets

    def _hist_job(self, acts, sg, loss, d_gray=None):
        """histogram losses on layers of the loss network: per-view losses into ``loss``, gradients (ReLU-masked) into
        the layers' entries of ``sg``.  ``d_gray`` [B,H,W,1]: the masked branch (styler_base.py:196-201) -- the density
        mask, bicubic-resized to each layer, removes its zero pixels from the source of the match.

        Deliberate divergence from the mounted reference (documented in DESIGN.md section 5, INTEGRATION.md and the
        hist.hip header): the template of a layer is the style image's features OF THAT LAYER and the weight is
        ``w_hist_layer`` -- the reference's lines 203 / 207 read the stale ``style_feature`` / ``w_style_layer`` left
        over from the style loop (the last style layer's placeholder and weight), which only coincide with this when
        hist_layer == [style_layer[-1]] and w_hist_layer == [w_style_layer[-1]]."""
        for name, wl in zip(self.hist_layers, self.w_hist_layers):
            if "input" in name:
                continue
            assert self.hist_targets is not None, "call set_hist_image first"
            F, padded = _logical(acts, name)
            m = None if d_gray is None else ops.resize_bicubic_tf1(d_gray.contiguous(), F.shape[1], F.shape[2])
            if padded:
                g = torch.zeros_like(F)
            else:
                g = sg.get(name)
                if g is None:
                    g = sg[name] = torch.zeros_like(F)
            ops.hist_loss(F, self.hist_targets[name], wl * self.w_hist, loss, g, relu_mask=_post_relu(name), mask=m)
            if padded:
                _add_logical(sg, name, acts[name], g)

    def _hist_input(self, dimg, loss, g_x, d_gray=None):
        """hist_layer 'input': the term on d_img itself (gradient wrt d_img = gradient wrt the mean-subtracted x)"""
        for name, wl in zip(self.hist_layers, self.w_hist_layers):
            if "input" in name:
                assert self.hist_targets is not None, "call set_hist_image first"
                m = None if d_gray is None else ops.resize_bicubic_tf1(d_gray.contiguous(), dimg.shape[1], dimg.shape[2])
                ops.hist_loss(dimg, self.hist_targets[name], wl * self.w_hist, loss, g_x, relu_mask=False, mask=m)

    def set_content_image(self, content_img, top_k=0):
        """content_img: float32 [h,w,3] in 0..255 at the loss-net input size, or None (styler_base.py:232-247);
        ``top_k`` > 0 (with the classifier logits as content layer): only the k largest |logits| of every row of the
        fetched feature survive, the others are zeroed (styler_base.py:240-245)"""
        if content_img is None or self.content_layer is None:
            self.content_feature = None
            return None
        dev = self.net.device
        s = torch.as_tensor(np.asarray(content_img, np.float32)).to(dev)
        mean = torch.tensor([0.485 * 255, 0.456 * 255, 0.406 * 255], dtype=torch.float32, device=dev)
        acts = self.net.forward((s - mean).unsqueeze(0).contiguous(), self.content_layer)
        cf = _logical(acts, self.content_layer)[0].clone()
        if top_k and top_k > 0:
            if "softmax2_pre_activation" not in self.content_layer:
                raise AssertionError("top_k > 0 needs content_layer softmax2_pre_activation (styler_base.py:241)")
            rows = cf.reshape(-1, cf.shape[-1])
            keep = torch.zeros_like(rows, dtype=torch.bool)
            keep.scatter_(1, rows.abs().topk(int(top_k), dim=1).indices, True)
            cf = (rows * keep).reshape(cf.shape).contiguous()
        self.content_feature = cf
        return self.content_feature

    # -- style targets (styler_base.py:249-278: the style image enters at d_img) -------------
    def out_hw(self, H, W):
        if np.isclose(self.resize_scale, 1):
            return H, W
        # int(float32(scale) * float32(dim)) as the TF graph computes it (styler_base.py:36-37)
        return int(np.float32(self.resize_scale) * np.float32(H)), int(np.float32(self.resize_scale) * np.float32(W))

    def set_style_image(self, style_img):
        """style_img: float32 [h,w,3] in 0..255 already at the loss-net input size"""
        dev = self.net.device
        s = torch.as_tensor(np.asarray(style_img, np.float32)).to(dev)
        mean = torch.tensor([0.485 * 255, 0.456 * 255, 0.406 * 255], dtype=torch.float32, device=dev)
        x = (s - mean).unsqueeze(0).contiguous()
        acts = self.net.forward(x, self.top)
        self.style_grams = {}
        for name in self.layers:
            F = acts[name]
            _, h, w, _ = F.shape
            self.style_grams[name] = ops.gram_fwd(F, 1.0 / (2.0 * h * w * _channels(acts, name, F)))
        return self.style_grams

    # -- forward only (rendered image, used for the final inference) ------------------------
    def _coef_form(self, V, shape):
        """transmittance rendering of a shape the segmented forward takes: the adjoint runs without the render-adjoint
        pass over the rotated volume (K4a; ``NFS_RENDER_COEF=0`` keeps that pass)"""
        return (self.mode == 0 and os.environ.get("NFS_RENDER_COEF", "1") != "0"
                and ops.render_coef_layout(V, *shape) is not None)

    def render(self, d, rot, keep_rotated=False, normalise=True):
        self.d_rot = None
        self._seg = None
        if self.rotate:
            if keep_rotated:
                self.d_rot = torch.empty((rot.shape[0],) + tuple(d.shape), dtype=torch.float32, device=d.device)
            if self.mode >= 2:                       # max / mean: rotate, then the ray reduction on the kept volume
                self.d_rot = ops.rotate_fwd(d.unsqueeze(-1), rot).squeeze(-1)
                img, rs = ops.render_fwd(self.d_rot, self.tau, self.mode)
                if not keep_rotated:
                    self.d_rot = None
            elif keep_rotated and self._coef_form(rot.shape[0], d.shape):
                # the kept volume holds u, not the samples: the rotate adjoint forms the sample gradient from it
                img, rs, _, self._seg = ops.rotate_render_fwd_coef(d, rot, self.tau, u_rot=self.d_rot)
            else:
                img, rs = ops.rotate_render_fwd(d, rot, self.tau, self.mode, d_rot=self.d_rot)
        else:
            img, rs = ops.render_fwd(d.unsqueeze(0), self.tau, self.mode)
        gmax = None
        V = img.shape[0]
        if self.liquid:
            norm = img
        elif normalise:
            norm, gmax = ops.maxnorm_fwd(img, max(V // self.v_batch, 1))
        else:
            norm = None                              # (the caller normalises: _chain's fused max-norm + loss-net input)
        return img, rs, norm, gmax

    def d_img(self, d, rot):
        """the 0..255 3-channel image the loss net sees (``self.d_img`` of the reference)"""
        _, _, norm, _ = self.render(d, rot)
        V, H, W = norm.shape
        H2, W2 = self.out_hw(H, W)
        dimg, _ = ops.loss_net_input_fwd(norm.unsqueeze(-1), H2, W2, want_x=False)
        return dimg

    def _gram_job(self, name, F, loss, unmasked=None, channels=None):
        """Gram matrix, style loss and the Gram gradient dF of one style layer.  dF carries the layer's ReLU mask unless
        the data gradient that will add it applies that mask anyway (then the name is recorded in ``unmasked`` and the
        GEMM epilogue does not read F a second time)."""
        wl = self.w_layers[self.layers.index(name)]
        _, h, w, c = F.shape
        scale = 1.0 / (2.0 * h * w * (channels or c))
        G = ops.gram_fwd(F, scale)
        Dm = ops.style_loss_fwd(G, self.style_grams[name], wl * self.w_style, loss)
        defer = unmasked is not None and self._defers_mask(name, F)
        if defer:
            unmasked.add(name)
        return ops.gram_bwd(F, Dm, scale, relu_mask=_post_relu(name) and not defer)

    def _defers_mask(self, name, F):
        """True when dF of style layer ``name`` is handed over WITHOUT its ReLU mask: the data gradient that adds it
        applies that very mask to the sum anyway ("Deferred ReLU mask")"""
        return (name != self.top and name != self.content_layer and os.environ.get("NFS_NO_DEFER_MASK") is None
                and self.net.masks_addend_of(name, F.shape))

    def _vgg_loss_grad(self, x, total_out=None):
        """x [B,h,w,3] (mean-subtracted) -> (dL/dx, per-image losses [B] of the style / content / histogram terms).

        Default: the Gram work of ALL style layers runs after the forward pass as three launches (tile pairs of every
        layer, slab reduction with the style loss folded in, the five Gram gradients as one batched GEMM launch:
        ``ops.gram_style_group``); the loss leaves the kernels as per-block partial sums (no atomics) and is summed
        here.  ``NFS_GRAM_GROUP=0`` restores the per-layer chain (Gram -> loss -> dF per layer), which can run on a
        side stream as soon as a layer exists (``gram_side_stream``)."""
        sg = {}
        unmasked = set()                            # style layers whose dF is handed over without its ReLU mask
        if not self.layers:                          # content / histogram terms only (run.bat:14-20: w_style 0)
            acts = self.net.forward(x, self.top, keep=self._keep())
            loss = torch.zeros(x.shape[0], dtype=torch.float32, device=x.device)
            self._content_job(acts, sg, loss)
            self._hist_job(acts, sg, loss)
            return self.net.backward(acts, sg, self.top, unmasked=unmasked), loss
        if self.gram_grouped:
            acts = self.net.forward(x, self.top, keep=self._keep())
            Fs = [acts[n] for n in self.layers]
            masks = []
            for n, F in zip(self.layers, Fs):
                d = self._defers_mask(n, F)
                if d:
                    unmasked.add(n)
                masks.append(_post_relu(n) and not d)
            parts, dFs, _ = ops.gram_style_group(Fs, [self.style_grams[n] for n in self.layers],
                                                 [w * self.w_style for w in self.w_layers], masks,
                                                 channels=[_channels(acts, n, F) for n, F in zip(self.layers, Fs)])
            sg.update(zip(self.layers, dFs))
            if total_out is not None:                # (sums_total(): no other loss term) the total straight from the
                torch.sum(parts.view(-1), dim=0, keepdim=True, out=total_out)     # per-block partial sums: one launch
                return self.net.backward(acts, sg, self.top, unmasked=unmasked), None
            loss = parts.sum(0)
            self._content_job(acts, sg, loss)
            self._hist_job(acts, sg, loss)
            return self.net.backward(acts, sg, self.top, unmasked=unmasked), loss
        loss = torch.zeros(x.shape[0], dtype=torch.float32, device=x.device)
        if not self.gram_side_stream:
            acts = self.net.forward(x, self.top, keep=self._keep())
            for name in self.layers:
                sg[name] = self._gram_job(name, acts[name], loss, unmasked, _channels(acts, name, acts[name]))
            self._content_job(acts, sg, loss)
            self._hist_job(acts, sg, loss)
            return self.net.backward(acts, sg, self.top, unmasked=unmasked), loss
        main = torch.cuda.current_stream(x.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=x.device)
        side = self._side
        side.wait_stream(main)                      # style targets were produced on the main stream

        top_inline = os.environ.get("NFS_GRAM_TOP_INLINE", "1") != "0"

        def on_layer(name, F):
            if name not in self.layers:
                return
            if name == self.top and top_inline:
                # nothing follows the top layer on the main stream: its Gram work is the critical path, run it in
                # place (no event hand-over to the side stream and back)
                sg[name] = self._gram_job(name, F, loss, unmasked)
                return
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            capturing = torch.cuda.is_current_stream_capturing()
            if not capturing:                       # a captured step owns its buffers: nothing to tell the allocator
                F.record_stream(side)
            with torch.cuda.stream(side):
                sg[name] = self._gram_job(name, F, loss, unmasked)
            if not capturing:
                sg[name].record_stream(main)

        acts = self.net.forward(x, self.top, on_layer=on_layer, keep=self._keep())
        main.wait_stream(side)
        self._content_job(acts, sg, loss)
        self._hist_job(acts, sg, loss)
        return self.net.backward(acts, sg, self.top, unmasked=unmasked), loss

    def _batch_views(self, V):
        """views per loss-net batch of the reference graph: v_batch (RenderStyleLoss: a property of the run, not of
        the shard -- a rank holding fewer views still divides by v_batch); the whole image batch for the 2-D loss
        (``v_batch is None``: one sess.run sees every image)"""
        return float(V) if self.v_batch is None else float(max(int(self.v_batch), 1))

    def _keep(self):
        """the activations the loss itself reads: the forward pass need not materialise the full-resolution output
        of a pooled layer that is not among them"""
        return (set(self.layers) | ({self.content_layer} if self.content_layer else set())
                | set(n for n in getattr(self, "hist_layers", []) if "input" not in n))

    def _content_job(self, acts, sg, loss):
        """adds the content term's per-view losses into ``loss`` and its gradient into the layer's entry of ``sg``"""
        if self.content_layer is None:
            return
        F, padded = _logical(acts, self.content_layer)
        V = F.shape[0]
        # the reference's means run over one loss-net batch (v_batch views); here all V views share the batch
        w = self.w_content * V / self._batch_views(V)
        if padded:
            g = torch.zeros_like(F)
        else:
            g = sg.get(self.content_layer)
            if g is None:
                g = sg[self.content_layer] = torch.zeros_like(F)
        ops.content_loss(F, w, loss, g, channel=self.content_channel, target=self.content_feature,
                         amp=self.w_content_amp, signed=not _post_relu(self.content_layer))
        if padded:
            _add_logical(sg, self.content_layer, acts[self.content_layer], g)

    # -- the hot step -----------------------------------------------------------------------
    accepts_live = True       # loss_and_grad(..., live=, dilate=): see ops.rotate_bwd_coef

    def _chain(self, d, rot, g_d, overwrite=False, total_out=None, live=None, dilate=1):
        """render -> loss net -> Gram losses -> full adjoint for the views ``rot`` on the CURRENT stream;
        g_d [D,H,W] += dL/dd (= with ``overwrite``, two-pass rotate adjoint only); returns the per-view losses"""
        D, H, W = d.shape
        H2, W2 = self.out_hw(H, W)
        hist_in = any("input" in n for n in self.hist_layers)
        # smoke render at the loss net's own size with nothing reading d_img: max-normalisation and the loss-net input
        # in one pass each way (same arithmetic; two launches and the [V,H,W] intermediates less per direction)
        fused_in = (not self.liquid and (H2, W2) == (H, W) and not (self.w_tv > 0 or hist_in)
                    and os.environ.get("NFS_FUSE_INPUT", "1") != "0")
        img, rs, norm, gmax = self.render(d, rot, keep_rotated=self.two_pass_adjoint, normalise=not fused_in)
        d_rot, seg = self.d_rot, self._seg
        self.d_rot = self._seg = None
        V = img.shape[0]
        if fused_in:
            dimg = None
            x, gmax = ops.maxnorm_input_fwd(img, max(V // self.v_batch, 1))
        else:
            dimg, x = ops.loss_net_input_fwd(norm.unsqueeze(-1), H2, W2, want_d_img=self.w_tv > 0 or hist_in)
        g_x, loss = self._vgg_loss_grad(x, total_out=total_out)
        if hist_in:
            self._hist_input(dimg, loss, g_x)
        if self.w_tv > 0:
            # the reference's batch mean runs over one loss-net batch (v_batch views, styler_base.py:211-213); with
            # all V local views in one batch the weight is rescaled so that the term is w_tv * sum_v TV_v / v_batch
            # whatever the number of views this rank holds -- ALSO when a rank holds fewer views than one loss-net
            # batch (8 views over 8 ranks with v_batch 2): the divisor is v_batch, never the local V, so that sharded
            # runs sum to the single-rank value
            tv = torch.zeros(1, dtype=torch.float32, device=d.device)
            ops.tv_loss(dimg, self.w_tv * V / self._batch_views(V), tv, g_x)
            loss = loss + tv / V
        if fused_in:
            g_img = ops.maxnorm_input_bwd(img, gmax, g_x)
        else:
            g_norm = ops.loss_net_input_bwd(g_x, H, W, 1).reshape(V, H, W)
            g_img = g_norm if self.liquid else ops.maxnorm_bwd(img, gmax, g_norm)
        if self.rotate and d_rot is not None and seg is not None:
            # u form: per-(ray, segment) coefficients from the image gradient, then the rotate adjoint alone
            ab, bounds = ops.render_ray_coef(g_img, seg, self.tau)
            ops.rotate_bwd_coef(d_rot, ab, rot, bounds, g_d_acc=g_d, overwrite=overwrite, live=live, dilate=dilate)
        elif self.rotate and d_rot is not None:
            # two-pass adjoint: streaming render adjoint on the kept rotated volume (re-using its
            # buffer for the per-sample gradient) + LDS-tiled output-stationary rotate adjoint
            g_rot, g_max = ops.render_bwd(d_rot, rs, g_img, self.tau, self.mode, g_d=d_rot, want_max=True)
            ops.rotate_bwd(g_rot.unsqueeze(-1), rot, g_d_acc=g_d.unsqueeze(-1), g_max=g_max, overwrite=overwrite)
        elif self.rotate:
            assert self.mode < 2, "the max / mean ray modes use the two-pass adjoint"
            ops.rotate_render_bwd(d, rot, rs, g_img, self.tau, self.mode, g_d_acc=g_d)
        elif overwrite:
            # one unrotated view: the render adjoint IS dL/dd -- written straight into g_d (no zero fill, no add pass)
            ops.render_bwd(d.unsqueeze(0), rs, g_img, self.tau, self.mode, g_d=g_d.unsqueeze(0))
        else:
            g_d.add_(ops.render_bwd(d.unsqueeze(0), rs, g_img, self.tau, self.mode)[0])
        return loss

    def writes_gradient(self, V):
        """True when loss_and_grad(..., overwrite=True) can write g_d without a zero fill: one view batch through the
        tiled rotate adjoint, whose tiles partition the volume -- or the single unrotated view, whose render adjoint is
        the gradient itself"""
        if not self.rotate:
            return True
        ngroups = self.view_groups if (V >= 4 and self.rotate and self.v_batch == 1) else 1
        return bool(self.two_pass_adjoint and min(ngroups, V) <= 1)

    def sums_total(self, V):
        """True when loss_and_grad(..., total_out=slot) can write the TOTAL loss of the view batch into ``slot`` itself (and
        return None instead of the per-view losses): the grouped Gram chain is the only loss term and the batch is one
        chain -- the total is then one reduction of the kernels' partial sums instead of a per-view reduction here and a
        second one over the views at the caller"""
        ngroups = self.view_groups if (V >= 4 and self.rotate and self.v_batch == 1) else 1
        return bool(self.gram_grouped and self.layers and not self.w_tv and not self.hist_layers
                    and not self.content_layer and min(ngroups, V) <= 1)

    def loss_and_grad(self, d, rot, g_d, overwrite=False, total_out=None, live=None, dilate=1):
        """d [D,H,W] (output of smooth3d_relu), rot [V,3,3] device tensor, g_d [D,H,W] += dL/dd.
        Returns loss per view [V] (device tensor).

        ``live`` (``ops.live_mask`` written by the forward advect; velocity variable only): g_d is needed only within
        ``dilate`` cells of a live voxel -- everywhere else the caller multiplies it by an exact zero -- and the rotate
        adjoint of the coefficient form leaves it unsummed there (``ops.rotate_bwd_coef``).

        With >= 4 local views (and per-view normalisation, v_batch == 1) the views are split into groups
        whose whole chains run on separate HIP streams: the MFMA-bound conv launches of one group overlap
        the VALU/LDS-bound render and rotate-adjoint kernels and the fill/drain phases of the other."""
        assert self.style_grams is not None or not self.layers, "call set_style_image first"
        V = rot.shape[0] if self.rotate else 1
        ngroups = self.view_groups if (V >= 4 and self.rotate and self.v_batch == 1) else 1
        ngroups = min(ngroups, V)
        if ngroups <= 1:
            return self._chain(d, rot, g_d, overwrite=overwrite and self.writes_gradient(V),
                               total_out=total_out if (total_out is not None and self.sums_total(V)) else None,
                               live=live, dilate=dilate)
        assert not overwrite, "overwrite needs a single view batch (see writes_gradient)"
        assert total_out is None, "total_out needs a single view batch (see sums_total)"
        nst = min(self.vgg_streams, ngroups)
        main = torch.cuda.current_stream(d.device)
        if len(self._streams) < nst:
            self._streams = [torch.cuda.Stream(d.device) for _ in range(nst)]
        loss = torch.empty(V, dtype=torch.float32, device=d.device)
        g_parts = [torch.zeros_like(g_d) for _ in range(nst)]     # one accumulator per stream (plain RMW inside)
        bounds = [V * i // ngroups for i in range(ngroups + 1)]
        for si in range(nst):
            self._streams[si].wait_stream(main)
        for gi in range(ngroups):
            si = gi % nst
            with torch.cuda.stream(self._streams[si]):
                lo, hi = bounds[gi], bounds[gi + 1]
                loss[lo:hi].copy_(self._chain(d, rot[lo:hi].contiguous(), g_parts[si], live=live, dilate=dilate))
        for si in range(nst):
            main.wait_stream(self._streams[si])
        for gp in g_parts:
            ops.axpy(g_d, gp, 1.0)
        return loss


def loss_capture_key(L):
    """what a captured graph of ``L.loss_and_grad`` has baked in by address or by value (style / content / histogram
    targets are device tensors, the loss weights are kernel arguments)"""
    grams = tuple(int(t.data_ptr()) for t in (L.style_grams or {}).values())
    cf = getattr(L, "content_feature", None)
    hyper = tuple(getattr(L, a, None) for a in ("w_style", "tau", "mode", "resize_scale", "rotate", "w_tv",
                                                "v_batch", "w_content", "content_layer", "content_channel",
                                                "w_content_amp"))
    # histogram term: template tensors by address, its weights and layer list by value
    hist = (tuple(sorted((n, int(t.data_ptr())) for n, t in (getattr(L, "hist_targets", None) or {}).items())),
            getattr(L, "w_hist", 0.0), tuple(getattr(L, "hist_layers", ())), tuple(getattr(L, "w_hist_layers", ())),
            getattr(L, "style_mask", False))
    return (grams, tuple(getattr(L, "layers", ())), tuple(getattr(L, "w_layers", ())), hyper,
            None if cf is None else int(cf.data_ptr()), hist, id(getattr(L, "net", None)))


class GraphedLoss(object):
    """``loss.loss_and_grad(d, rot, g_d)`` of a few views, replayed as ONE hipGraph where the host cannot keep up with
    it.  A one-view chain is 60-110 small launches: on a box with a slow host the host needs 1.1-1.4 ms to issue what
    the GPU runs in 0.58-0.97 (tools/loss_chain_host.py: the chocolate-scale VGG chain 1.39 -> 0.58 ms, the reference
    smokegun configuration on Inception-v1 1.11 -> 0.97 ms); on a box whose host keeps up, eager submission is the
    faster of the two by ~10 % (graph nodes do not overlap their launch latencies the way back-to-back stream packets
    do).  Hosts differ by 2x between boxes of one pool, so the choice is MEASURED: call 1 runs eagerly (lazy state:
    packed filters, tile tuner, workspaces); call 2 eagerly between two synchronisations, timing both what the host
    needed to issue it and when the GPU finished -- if the GPU finished as soon as the host did (issue time > 85 % of the
    wall time; calls 2-4 are timed this way and the MEDIAN ratio decides, so that one-off lazy work in one of them does
    not) the chain is host-bound and calls 5.. are replays of a captured graph, otherwise it stays eager (``mode``
    = 'graph' | 'eager'; ``force`` = True / False skips the trial: ``NFS_GRAPH=0/1`` is the reproducible setting, the
    measured choice can differ between hosts and runs).  A change of anything the capture bakes in
    (``loss_capture_key``, shapes) starts over.  Inputs are copied into static buffers; the returned loss and gradient
    tensors are valid until the next call."""

    def __init__(self, loss, force=None):
        self.loss = loss
        self.force = force
        self._reset()

    def _reset(self):
        self._graph = None
        self._key = None
        self._calls = 0
        self.mode = None if self.force is None else ("graph" if self.force else "eager")
        self.trial = None
        self._trials = []

    def _eager(self, d, rot):
        if isinstance(self.loss, ImageStyleLoss):        # (d, d_gray) -> (losses, gradient)
            return self.loss.loss_and_grad(d, rot)
        V = 1 if rot is None else int(rot.shape[0])
        if hasattr(self.loss, "writes_gradient") and self.loss.writes_gradient(V):
            g_d = torch.empty_like(d)
            return self.loss.loss_and_grad(d, rot, g_d, overwrite=True), g_d
        g_d = torch.zeros_like(d)
        return self.loss.loss_and_grad(d, rot, g_d), g_d

    def __call__(self, d, rot):
        import time
        key = (loss_capture_key(self.loss), tuple(d.shape), None if rot is None else tuple(rot.shape))
        if self._key is not None and key != self._key:
            self._reset()
        self._key = key
        self._calls += 1
        if self.mode == "eager" or self._calls == 1:
            return self._eager(d, rot)
        if self.mode is None:
            # three timed eager calls (2, 3, 4), the median ratio decides: one trial can catch one-off lazy work
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = self._eager(d, rot)
            t_issue = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_wall = time.perf_counter() - t0
            self._trials.append((t_issue, t_wall))
            if len(self._trials) == 3:
                self.trial = sorted(self._trials, key=lambda tw: tw[0] / tw[1])[1]
                self.mode = "graph" if self.trial[0] > 0.85 * self.trial[1] else "eager"
                if os.environ.get("NFS_VERBOSE"):
                    print("GraphedLoss: %s (issue %.3f ms of %.3f ms wall, median of 3; NFS_GRAPH=0/1 fixes the choice)"
                          % (self.mode, 1e3 * self.trial[0], 1e3 * self.trial[1]))
            return out
        if self._graph is None:
            self._d = d.clone()
            self._rot = None if rot is None else rot.clone()
            self._gd = torch.zeros_like(d)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                if isinstance(self.loss, ImageStyleLoss):
                    self._losses, self._gd = self.loss.loss_and_grad(self._d, self._rot)
                elif hasattr(self.loss, "writes_gradient") and self.loss.writes_gradient(
                        1 if self._rot is None else int(self._rot.shape[0])):
                    self._losses = self.loss.loss_and_grad(self._d, self._rot, self._gd, overwrite=True)
                else:
                    self._gd.zero_()
                    self._losses = self.loss.loss_and_grad(self._d, self._rot, self._gd)
            self._graph = g
        else:
            self._d.copy_(d)
            if rot is not None and rot.data_ptr() != self._rot.data_ptr():
                self._rot.copy_(rot)
        self._graph.replay()
        return self._losses, self._gd


class argparse_ns(object):
    """plain attribute bag"""


class TFAdamState(object):
    """tf.compat.v1.train.AdamOptimizer state (m, v, beta powers) living on the device; one
    instance per ``opt_id`` (styler_3p.py:315-323), persisting across frames/views/octaves."""

    def __init__(self, beta1=0.9, beta2=0.999, eps=1e-8):
        self.b1, self.b2, self.eps = np.float32(beta1), np.float32(beta2), np.float32(eps)
        self.b1p, self.b2p = np.float32(1.0), np.float32(1.0)
        self.m = None
        self.v = None
        # OR of every live mask since m and v were zeroed (step_through_advect with a mask): voxels outside it still have
        # m = v = +0.  Usable only while nothing but that kernel has written m or v (their version counters say)
        self._ever = None
        self._ever_key = None

    def ever_mask(self):
        """the mask above, or None when m or v have been written by anything else since they were zeroed (as seen by their
        torch version counters: a write that bypasses them -- through ``.data``, or a kernel on ``m.data_ptr()`` -- is not
        seen; set ``_ever = None`` after one)"""
        if self._ever is None or self.m is None or self._ever_key != (self.m._version, self.v._version):
            self._ever = None
            return None
        return self._ever

    def step(self, x, g, lr):
        if self.m is None or self.m.shape != x.shape:
            self.m = torch.zeros_like(x)
            self.v = torch.zeros_like(x)
        # beta powers are float32 variables multiplied once per step in TF
        self.b1p = np.float32(self.b1p * self.b1)
        self.b2p = np.float32(self.b2p * self.b2)
        lr_t = np.float32(lr) * np.sqrt(np.float32(1.0) - self.b2p) / (np.float32(1.0) - self.b1p)
        ops.adam_tf_step(x, self.m, self.v, g, float(lr_t), float(self.b1), float(self.b2), float(self.eps))

    def step_through_advect(self, vel, d0, g_adv, lr, adv_next=None, live_next=None, live_current=False):
        """the same update for the velocity variable of ``advect(d0, vel)`` given dL/d(advected density): the
        velocity gradient is formed and consumed inside one kernel (never written to HBM).  ``adv_next`` [D,H,W]
        (optional) receives advect(d0, updated vel) -- the next iteration's forward sample -- in the same pass, ``live_next``
        its live mask.  ``live_current``: ``live_next`` holds the mask of the CURRENT sample on entry (and ``adv_next`` that
        sample) -- the waves whose voxels never were live are then left out (``ever_mask``)"""
        fresh = self.m is None or self.m.shape != vel.shape
        if fresh:
            self.m = torch.zeros_like(vel)
            self.v = torch.zeros_like(vel)
            self._ever = None
        if fresh and live_next is not None:
            self._ever = torch.zeros_like(live_next)
            self._ever_key = (self.m._version, self.v._version)
        ever = self.ever_mask() if (live_current and live_next is not None and adv_next is not None) else None
        self.b1p = np.float32(self.b1p * self.b1)
        self.b2p = np.float32(self.b2p * self.b2)
        lr_t = np.float32(lr) * np.sqrt(np.float32(1.0) - self.b2p) / (np.float32(1.0) - self.b1p)
        tracked = self._ever is not None and self._ever_key == (self.m._version, self.v._version)
        ops.advect_bwd_adam(d0, vel, g_adv, self.m, self.v, float(lr_t), float(self.b1), float(self.b2),
                            float(self.eps), adv_next=adv_next, live_next=live_next, ever=ever)
        if ever is not None:
            self._ever_key = (self.m._version, self.v._version)      # (our own write: the mask stays usable)
        elif tracked:
            self._ever = None                                        # (a step without the mask update: it no longer covers m, v)


    def step_through_advect_slab(self, vel_slab, d0, g_adv_slab, z0, lr, adv_next=None):
        """``step_through_advect`` on the planes [z0, z0 + nz) only: the moments exist for these planes alone (the Adam
        state of a D-slab-sharded run is sharded with the variable)"""
        if self.m is None or self.m.shape != vel_slab.shape:
            self.m = torch.zeros_like(vel_slab)
            self.v = torch.zeros_like(vel_slab)
        self.b1p = np.float32(self.b1p * self.b1)
        self.b2p = np.float32(self.b2p * self.b2)
        lr_t = np.float32(lr) * np.sqrt(np.float32(1.0) - self.b2p) / (np.float32(1.0) - self.b1p)
        ops.advect_bwd_adam_slab(d0, vel_slab, g_adv_slab, self.m, self.v, z0, float(lr_t), float(self.b1),
                                 float(self.b2), float(self.eps), adv_next=adv_next)


class LBFGSState(object):
    """Limited-memory BFGS for the outer loop (north_star names L-BFGS beside Adam; the mounted reference only has Adam):
    one quasi-Newton step per ``step(x, g, lr)`` from the gradient the HIP chain delivers -- two-loop recursion over the
    last ``history`` (s, y) pairs, initial Hessian scale y.s / y.y, a pair kept only when y.s > 1e-10, fixed step
    ``lr`` (the first step min(1, 1/|g|_1) lr), no line search: the arithmetic of ``torch.optim.LBFGS(lr, max_iter=1,
    history_size=history, line_search_fn=None)`` called once per iteration, against which tests/test_optim_cpu.py holds
    it.  Everything stays on the variable's device (elementwise torch kernels and reductions); same call surface as
    ``TFAdamState.step``, so the stylizers take either."""

    def __init__(self, history=10, tolerance_grad=1e-7, tolerance_change=1e-9):
        self.history = int(history)
        # torch.optim.LBFGS's stopping tests, applied per call: nothing moves when max|g| <= tolerance_grad (an empty or
        # fully masked frame has g = 0: the first step's 1/|g|_1 would divide by zero) or when the quasi-Newton direction
        # is not a descent direction (g.d > -tolerance_change)
        self.tol_g, self.tol_c = float(tolerance_grad), float(tolerance_change)
        self.n = 0
        self.S, self.Y, self.ro = [], [], []
        self.d = self.prev_g = None
        self.t = None
        self.H = 1.0

    def step(self, x, g, lr):
        g = g.reshape(-1)
        if self.prev_g is not None and self.prev_g.numel() != g.numel():
            raise ValueError("LBFGSState: the variable changed size (%d -> %d elements): the curvature history belongs to "
                             "ONE variable -- use one state per variable (engine.optimizer_slot)"
                             % (self.prev_g.numel(), g.numel()))
        if float(g.abs().max()) <= self.tol_g if g.numel() else True:
            return
        self.n += 1
        if self.n == 1:
            d = g.neg()
            self.H = 1.0
        else:
            y = g - self.prev_g
            s_ = self.d * self.t
            ys = float(y.dot(s_))                          # (dot products, as torch.optim.LBFGS forms them: a sum of
            # products in another order can land on the other side of the 1e-10 curvature test)
            if ys > 1e-10:
                if len(self.S) == self.history:
                    self.S.pop(0); self.Y.pop(0); self.ro.pop(0)
                self.S.append(s_); self.Y.append(y); self.ro.append(1.0 / ys)
                self.H = ys / float(y.dot(y))
            q = g.neg()
            al = [0.0] * len(self.S)
            for i in range(len(self.S) - 1, -1, -1):
                al[i] = float(self.S[i].dot(q)) * self.ro[i]
                q.add_(self.Y[i], alpha=-al[i])
            d = q.mul(self.H)
            for i in range(len(self.S)):
                be = float(self.Y[i].dot(d)) * self.ro[i]
                d.add_(self.S[i], alpha=al[i] - be)
        self.prev_g = g.clone()
        self.t = min(1.0, 1.0 / float(g.abs().sum())) * float(lr) if self.n == 1 else float(lr)
        self.d = d
        if float(g.dot(d)) > -self.tol_c:
            self.t = 0.0                 # (no move; the pair of the next call is then s = 0: y.s = 0, not kept)
            return
        x.reshape(-1).add_(d, alpha=self.t)


def optimizer_slot(kind, t, frames_per_opt):
    """key of the optimiser state frame ``t`` steps with.  Adam: ``t // frames_per_opt`` -- one (m, v, step count) shared
    by the frames of a group, updated by them in turn (styler_3p.py:315-323; the moments are running averages and
    tolerate it).  L-BFGS: one state PER FRAME -- its (s, y) pairs are differences of consecutive gradients of one
    variable; pairs mixed across frames (or across particle counts) are not curvature of anything."""
    return ("lbfgs", int(t)) if kind == "lbfgs" else int(t) // int(frames_per_opt)


def make_optimizer(kind="adam"):
    """one optimiser state per ``opt_id`` (styler_3p.py:315-323): TF ApplyAdam (the reference) or L-BFGS (config.optimizer)"""
    if kind in (None, "", "adam"):
        return TFAdamState()
    if kind == "lbfgs":
        return LBFGSState()
    raise ValueError("optimizer %r: 'adam' or 'lbfgs'" % (kind,))


class GridStylizer(object):
    """TNST-style grid path assembled from the reference's operators:
        d^ = advect(d0, vel)  ->  smooth+max  ->  RenderStyleLoss
    with the velocity field ``vel`` [D,H,W,3] (target 'v') or the density itself (target 'd')
    as the Adam variable.  Views shard over ranks (``views=sum``): each rank evaluates its slice
    of the rotation matrices; the exchange is ONE all-reduce's worth of link traffic per iteration,
    split around the field work so that this work is sharded too (``slab`` mode, the default for the
    velocity variable): reduce-scatter of the density-field gradient over D-slabs -> slab-local smooth
    adjoint, advect adjoint + ApplyAdam, advect, smooth -> all-gather of the smoothed density (see
    ``_slab_setup``).  Otherwise the gradient is all-reduced and every rank applies the identical step.
    ``bind`` swaps the frame (density, variable, Adam state) under the same
    stylizer: the frame loop of a sequence (styler_grid.py) re-uses one instance."""

    def __init__(self, loss, d0, k=3, target="v", lr=0.1, process_group=None, graph=None, optimizer="adam"):
        self.loss = loss
        self.d0 = d0.contiguous()
        self.k = float(k)
        self.target = target
        self.lr = float(lr)
        self.pg = process_group
        self.adam = make_optimizer(optimizer)        # (named after the reference's optimiser; L-BFGS on request)
        # the advect adjoint consumed inside the Adam kernel: only when the optimiser IS Adam
        self.fuse_adam = os.environ.get("NFS_FUSE_ADAM", "1") != "0" and isinstance(self.adam, TFAdamState)
        # hipGraph replay of the forward + adjoint (about 130 launches a step; the host needs 1.25 ms to issue
        # them one by one, which is the whole step at one view per rank)
        # (measured: 200^3 x 8 views 3.98 -> 3.90 ms, 200^3 x 1 view 1.35 -> 1.41 ms, 100^3 x 1 view 1.20 -> 1.06 ms:
        # it pays where the step is shorter than the 1.15 ms the host needs to issue it).  None = decide at the first
        # step from the problem size; NFS_GRAPH=0/1 or the constructor argument override.
        env = os.environ.get("NFS_GRAPH")
        self.use_graph = (None if env is None else env == "1") if graph is None else bool(graph)
        self._graph = None
        self._graph_key = None
        self._graph_rot = None
        self._graph_total = None
        self._graph_warm = 0
        self._pending = None
        self._owns_buffers = False
        # advect(d0, var) of the NEXT iteration written by the Adam kernel of this one (ops.advect_bwd_adam(adv_next=...)):
        # the buffer, and what it was computed from (tensor identities + torch version counters; any in-place torch op on
        # the variable or the density, a bind() or a re-assignment makes it stale and the forward advect runs by itself)
        self.fuse_advect = os.environ.get("NFS_FUSE_ADVECT", "1") != "0"
        self._adv_buf = None
        self._adv_src = None
        # Dead-region skipping (velocity variable): the kernels that write the forward advect also write the mask of the
        # voxels whose back-traced density corners differ; everywhere else the advect adjoint multiplies the incoming
        # gradient by an exact zero, so step() / gradient() let the rotate adjoint skip what only feeds those voxels.
        # The update is bit-identical with it on and off (tests/test_dead_skip_gpu.py); NFS_DEAD_SKIP=0 turns it off.
        self.dead_skip = os.environ.get("NFS_DEAD_SKIP", "1") != "0"
        self._live_buf = None
        self._live_mark = None
        D, H, W = d0.shape
        if target == "v":
            self.var = torch.zeros(D, H, W, 3, dtype=torch.float32, device=d0.device)
        else:
            self.var = d0.clone()
        # field gradient and the summed loss share one buffer: the multi-rank exchange is ONE all-reduce
        # (the loss rides in the slot behind the gradient)
        n = D * H * W
        self.slab = None
        if self.pg is not None and os.environ.get("NFS_SLAB_SHARD", "1") != "0":
            self._slab_setup()
        if self.slab is None:
            self._gbuf = torch.zeros(n + 4, dtype=torch.float32, device=d0.device)
            self.g_ds = self._gbuf[:n].view(D, H, W)
            self._loss_slot = self._gbuf[n:n + 1]

    # ---- D-slab sharding of the replicated field work (view-sharded runs) ---------------------------------------------------
    def _slab_setup(self):
        """Strong scaling of the 8-view problem is capped by what every rank repeats: advect, smooth, their adjoints and
        ApplyAdam on the whole field (~0.2 ms of a 1.1 ms one-view step).  All of it is LOCAL in z up to a one-plane
        stencil, so the all-reduce of the density gradient is taken apart into its two halves with that work in between:

            reduce-scatter(g_ds)      rank k receives the summed gradient of ITS slab of planes, with a two-plane halo
                                      (the chunks of the send buffer overlap: no separate halo exchange)
            smooth adjoint            on slab +- 2 planes -> valid on slab +- 1
            advect adjoint + Adam     on slab +- 1: the variable, its moments and the update exist per slab (+ one plane
                                      each side, computed identically by both neighbours -- deterministic kernels)
            advect, smooth            next iteration's field on slab +- 1 -> valid smoothed density on the slab
            all-gather(d_s)           every rank holds the whole smoothed density for its views' rotate + render

        Same bytes over the links as the all-reduce it replaces (+ 4 halo planes per chunk); d0 stays replicated and
        constant (the back-traced points leave the slab).  The loss rides in an extra plane of every chunk.  Off when
        the variable is the density itself, when a slab would be thinner than two planes, or with NFS_SLAB_SHARD=0."""
        rank, world = parallel.rank_world(self.pg)
        D, H, W = self.d0.shape
        force = os.environ.get("NFS_SLAB_SHARD") == "2"           # also on a one-rank group (RCCL smoke test on one GPU)
        if (world <= 1 and not force) or self.target != "v" or not self.fuse_adam or D // world < 2 or (H * W) % 4 \
                or min(D, H, W) < 2:
            return
        dev = self.d0.device
        cs, plan = parallel.slab_plan(D, world)
        z0, z1 = plan[rank]
        sl = argparse_ns()
        sl.rank, sl.world, sl.cs, sl.z0, sl.z1 = rank, world, cs, z0, z1
        sl.lo, sl.hi = max(z0 - 1, 0), min(z1 + 1, D)                     # planes whose variable this rank maintains
        # padded gradient volume: planes [2, D + 2) = g_ds, two zero planes either side, one more for the loss
        self._gpad = torch.zeros(D + 5, H, W, dtype=torch.float32, device=dev)
        self.g_ds = self._gpad[2:D + 2]
        self._loss_slot = self._gpad[D + 4].view(-1)[:1]
        sl.idx = torch.tensor(parallel.slab_pack_index(D, world), dtype=torch.int64, device=dev)    # (NFS_SLAB_PACK=torch)
        sl.pack = torch.empty(world, cs + 5, H, W, dtype=torch.float32, device=dev)
        sl.recv = torch.empty(cs + 5, H, W, dtype=torch.float32, device=dev)
        # padded smoothed density: planes [2, 2 + world * cs) are gathered, d_s = planes [2, 2 + D)
        sl.ds_full = torch.zeros(world * cs + 4, H, W, dtype=torch.float32, device=dev)
        sl.stage = torch.zeros(cs, H, W, dtype=torch.float32, device=dev)
        self.slab = sl

    def gather_variable(self):
        """slab mode: every rank keeps only its planes of the variable current -- all-gather them into ``self.var``
        (results, checkpoints); a no-op otherwise"""
        sl = self.slab
        if sl is None:
            return self.var
        D, H, W = self.d0.shape
        mine = torch.zeros(sl.cs, H, W, 3, dtype=torch.float32, device=self.var.device)
        if sl.z1 > sl.z0:
            mine[:sl.z1 - sl.z0] = self.var[sl.z0:sl.z1]
        full = torch.empty(sl.world, sl.cs, H, W, 3, dtype=torch.float32, device=self.var.device)
        parallel.all_gather_into(full, mine, group=self.pg)
        self.var.copy_(full.view(sl.world * sl.cs, H, W, 3)[:D])
        return self.var

    def _forward_field_slab(self):
        sl = self.slab
        assert self._pending is None, "bind() is not available with the D-slab sharding (one frame per stylizer)"
        D, H, W = self.d0.shape
        if sl.z1 > sl.z0:
            self.d_adv = self._advect_now()
            d_s_ext = ops.smooth3d_relu_fwd(self.d_adv, self.k)        # exact on the slab (cut planes are one further out)
            own = d_s_ext[sl.z0 - sl.lo: sl.z0 - sl.lo + (sl.z1 - sl.z0)]
            if sl.z1 - sl.z0 == sl.cs:
                src = own
            else:
                sl.stage[:sl.z1 - sl.z0].copy_(own)
                src = sl.stage
        else:
            src = sl.stage                                             # an idle rank contributes zero planes
        parallel.all_gather_into(sl.ds_full[2:2 + sl.world * sl.cs].view(sl.world, sl.cs, H, W), src, group=self.pg)
        self.d_s = sl.ds_full[2:2 + D]
        return self.d_s

    def _step_slab(self, rot_local):
        """one iteration with the field work sharded over D-slabs (see ``_slab_setup``)"""
        sl = self.slab
        D, H, W = self.d0.shape
        if self.use_graph:
            self._forward_field_slab()                                  # (holds a collective: outside the graph)
            self._loss_gradient_graphed(rot_local)
        else:
            losses, _ = self.field_gradient(rot_local)
            torch.sum(losses, dim=0, keepdim=True, out=self._loss_slot)
        if os.environ.get("NFS_SLAB_PACK") == "torch":                  # (the gather by index table the copy kernel replaced)
            torch.index_select(self._gpad, 0, sl.idx, out=sl.pack.view(-1, H, W))
        else:
            ops.slab_pack(self._gpad, sl.pack, D, sl.world, sl.cs)
        parallel.reduce_scatter_sum(sl.recv, sl.pack, group=self.pg)
        total = sl.recv[sl.cs + 4].view(-1)[0].clone()
        if sl.z1 > sl.z0:
            # chunk planes = global [z0 - 2, z0 + cs + 2); the same planes of the padded smoothed density
            g_adv = ops.smooth3d_relu_bwd(sl.ds_full[sl.z0:sl.z0 + sl.cs + 4], sl.recv[:sl.cs + 4], self.k)
            off = sl.lo - (sl.z0 - 2)
            adv = self._adv_target()
            self.adam.step_through_advect_slab(self.var[sl.lo:sl.hi], self.d0, g_adv[off:off + (sl.hi - sl.lo)], sl.lo,
                                               self.lr, adv_next=adv)
            if adv is not None:
                self._adv_mark()
        else:
            self.adam.b1p = np.float32(self.adam.b1p * self.adam.b1)      # (an idle rank keeps the step count)
            self.adam.b2p = np.float32(self.adam.b2p * self.adam.b2)
        return total

    def bind(self, d0, var=None, adam=None):
        """Re-point the stylizer at another frame of the same shape (takes effect at the next step / gradient).
        Without hipGraph replay this is a pointer swap; a captured graph has the addresses of ``d0`` and ``var`` baked
        in, so there the stylizer keeps private buffers and the frame's data is copied into them."""
        assert tuple(d0.shape) == tuple(self.d0.shape)
        self._pending = (d0, var)
        if adam is not None:
            self.adam = adam
            self.fuse_adam = self.fuse_adam and isinstance(adam, TFAdamState)

    def _apply_binding(self):
        if self._pending is None:
            return
        d0, var = self._pending
        self._pending = None
        if self.use_graph:
            if not self._owns_buffers:
                self.d0, self.var = self.d0.clone(), self.var.clone()
                self._owns_buffers = True
            self.d0.copy_(d0)
            if var is not None:
                self.var.copy_(var.reshape(self.var.shape))
        else:
            self.d0 = d0.contiguous()
            if var is not None:
                self.var = var

    # ---- the forward advect of iteration i + 1 rides in the Adam kernel of iteration i ---------------------------------------
    def _adv_target(self):
        """the buffer the fused Adam kernel writes the next forward sample into (None: fusion off / not applicable)"""
        if not (self.fuse_advect and self._fused_step_ok()):
            return None
        sl = self.slab
        shape = tuple(self.d0.shape) if sl is None else (sl.hi - sl.lo,) + tuple(self.d0.shape[1:])
        if self._adv_buf is None or tuple(self._adv_buf.shape) != shape:
            self._adv_buf = torch.empty(shape, dtype=torch.float32, device=self.d0.device)
            self._adv_src = None
        return self._adv_buf

    def _fused_step_ok(self):
        """may step() consume the advect adjoint inside the Adam kernel (``nfs_advect_bwd_adam*``: volumes of at least two
        cells a side whose cell count is a multiple of 4)?  The stored forward advect exists only then: any other shape
        updates the variable through ``TFAdamState.step``, which writes no next forward sample."""
        D, H, W = self.d0.shape
        return self.target == "v" and self.fuse_adam and min(D, H, W) >= 2 and (D * H * W) % 4 == 0

    def _adv_mark(self, live=False):
        self._adv_src = (self.var, self.var._version, self.d0, self.d0._version)
        self._live_mark = self._adv_src if live else None

    def _live_target(self):
        """the mask buffer the advect kernels fill beside the stored forward advect (None: skipping off / not applicable:
        whole-volume velocity runs whose loss takes a mask; slab-sharded ranks hold only their planes' velocity)"""
        if not (self.dead_skip and self.slab is None and getattr(self.loss, "accepts_live", False)
                and self._adv_target() is not None):
            return None
        if self._live_buf is None or self._live_shape != tuple(self.d0.shape):
            self._live_buf = ops.live_mask(*self.d0.shape, like=self.d0)
            self._live_shape = tuple(self.d0.shape)
            self._live_mark = None
        return self._live_buf

    def _live_valid(self):
        return (self._live_buf is not None and self._live_mark is not None and self._live_mark is self._adv_src
                and self._adv_valid())

    def _live_kw(self):
        """keyword arguments for loss_and_grad when the mask of the CURRENT forward advect exists"""
        if self._live_target() is None or not self._live_valid():
            return {}
        return {"live": self._live_buf, "dilate": 1 if self.k > 0 else 0}

    def invalidate_forward(self):
        """forget the stored forward advect.  In-place torch ops on ``var`` / ``d0`` (also through ``detach()`` views, which
        share the version counter), ``bind()`` and re-assignment are noticed by themselves; a write that bypasses the
        counter -- through ``tensor.data``, or by a kernel called on ``var.data_ptr()`` directly -- is not: call this
        after one."""
        self._adv_src = None

    def _adv_valid(self):
        a = self._adv_src
        return (a is not None and self._adv_buf is not None and a[0] is self.var and a[1] == self.var._version
                and a[2] is self.d0 and a[3] == self.d0._version)

    def _advect_now(self):
        """d_adv = advect(d0, var) -- from the previous step's Adam kernel when it is still current, else computed here
        (into the same buffer: a captured graph reads it by address)"""
        sl = self.slab
        buf = self._adv_target()
        live = self._live_target()
        if buf is not None and self._adv_valid() and (live is None or self._live_valid()):
            return buf
        if sl is None:
            out = ops.advect_fwd(self.d0.unsqueeze(-1), self.var, out=None if buf is None else buf.unsqueeze(-1),
                                 live=live).squeeze(-1)
        else:
            out = ops.advect_fwd_slab(self.d0, self.var[sl.lo:sl.hi], sl.lo, out=buf)
        if buf is not None:
            self._adv_mark(live=live is not None)
        return out

    def forward_field(self):
        if self.slab is not None:
            return self._forward_field_slab()
        self._apply_binding()
        if self.target == "v":
            self.d_adv = self._advect_now()
        else:
            self.d_adv = self.var
        self.d_s = ops.smooth3d_relu_fwd(self.d_adv, self.k)
        return self.d_s

    def field_gradient(self, rot_local, total=False, for_variable=False):
        """forward + adjoint down to the smoothed density: (loss_per_view, dL/d d_s of the LOCAL views); ``total``: see
        _loss_gradient.  ``for_variable``: the caller only takes the gradient on to the variable (step(), gradient()) --
        with a velocity variable dL/d d_s is then left unsummed where the advect adjoint multiplies it by zero"""
        d_s = self.forward_field()
        return self._loss_gradient(d_s, rot_local, total=total, live_kw=self._live_kw() if for_variable else None)

    def _loss_gradient(self, d_s, rot_local, total=False, live_kw=None):
        """total: let the loss write the summed loss of the local views into the loss slot where it can (then None is
        returned instead of the per-view losses)"""
        V = rot_local.shape[0]
        fresh = hasattr(self.loss, "writes_gradient") and self.loss.writes_gradient(V)
        if not fresh:
            self.g_ds.zero_()
        kw = {"overwrite": True} if fresh else {}
        if live_kw:
            kw.update(live_kw)
        if total and hasattr(self.loss, "sums_total") and self.loss.sums_total(V):
            kw["total_out"] = self._loss_slot
        losses = self.loss.loss_and_grad(d_s, rot_local, self.g_ds, **kw)
        return losses, self.g_ds

    def variable_gradient(self, g_ds):
        """adjoint of smooth+max and advect: dL/d variable from dL/d d_s (deterministic kernels: every
        rank computes the identical result from the all-reduced g_ds)"""
        g_adv = ops.smooth3d_relu_bwd(self.d_s, g_ds, self.k)
        if self.target == "v":
            _, g_var = ops.advect_bwd(self.d0.unsqueeze(-1), self.var, g_adv.unsqueeze(-1), need_d=False,
                                      need_vel=True)
            return g_var
        return g_adv

    def gradient(self, rot_local):
        """one forward+backward over the local views; returns (loss_per_view, grad wrt variable)"""
        losses, g_ds = self.field_gradient(rot_local, for_variable=self.target == "v")
        return losses, self.variable_gradient(g_ds)

    def _capture_key(self, rot_local):
        """everything a captured graph has baked in by address or by value: a mismatch forces a re-capture
        (set_style_image / set_content_image / set_hist_image allocate new targets; loss weights are kernel
        arguments)"""
        return loss_capture_key(self.loss) + (int(self.d0.data_ptr()), int(self.var.data_ptr()),
                                              tuple(rot_local.shape), self.k, self.target)

    def _loss_gradient_graphed(self, rot_local):
        """the graph without the field ops (slab mode: their all-gather stays outside the capture)"""
        return self._field_gradient_graphed(rot_local, with_field=False)

    def _field_gradient_graphed(self, rot_local, with_field=True):
        """field_gradient as one hipGraph: the first call runs eagerly (lazy state: packed Winograd filters, side
        streams, workspaces), the second is captured, later ones replay.  The capture is keyed on what it bakes in
        (style / content targets, loss hyper-parameters, the addresses of d0 and the variable, the number of views):
        when any of that changes the graph is dropped and captured again.  The view matrices are copied into a
        static buffer."""
        body = ((lambda r: self.field_gradient(r, total=True, for_variable=self.target == "v")) if with_field
                else (lambda r: self._loss_gradient(self.d_s, r, total=True)))

        def to_slot(losses):              # (None: the loss chain has written the total into the slot itself)
            if losses is not None:
                torch.sum(losses, dim=0, keepdim=True, out=self._loss_slot)     # one kernel: reduce straight into the slot
        if with_field and self.target == "v" and self._adv_target() is not None:
            # the captured forward starts from the advected density in its fixed buffer: bring it up to date eagerly when
            # the previous step's Adam kernel has not left it there (first step, a re-bound frame, a variable set by hand)
            self._apply_binding()
            self._advect_now()
        key = self._capture_key(rot_local) + (with_field, self._adv_target() is not None,
                                              with_field and bool(self._live_kw()))
        if self._graph is not None and key != self._graph_key:
            self._graph = None
            self._graph_warm = 0
        if self._graph is None:
            if self._graph_warm < 1:
                self._graph_warm += 1
                losses, g_ds = body(rot_local)
                to_slot(losses)
                return self._loss_slot, g_ds
            self._graph_rot = rot_local.clone()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                losses, _ = body(self._graph_rot)
                to_slot(losses)
            self._graph = g
            self._graph_key = key
            self._graph_rot_src = (rot_local, rot_local._version)
        elif rot_local.data_ptr() != self._graph_rot.data_ptr():
            # (the same tensor object, not modified in place since it was copied: nothing to copy -- a launch per step)
            src = getattr(self, "_graph_rot_src", None)
            if src is None or src[0] is not rot_local or src[1] != rot_local._version:
                self._graph_rot.copy_(rot_local)
                self._graph_rot_src = (rot_local, rot_local._version)
        self._graph.replay()
        return self._loss_slot, self.g_ds

    def _loss_chain_host_bound(self, rot_local):
        """one extra evaluation of the loss chain between two synchronisations: True when the GPU finishes it as soon as
        the host has issued it (issue time > 85 % of the wall time), i.e. when a hipGraph replay would be faster"""
        import time
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        self._loss_gradient(self.d_s, rot_local)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_wall = time.perf_counter() - t0
        self.graph_trial = (t_issue, t_wall)
        return t_issue > 0.85 * t_wall

    def step(self, rot_local, loss_view=False):
        """one iteration: loss + gradient of the local views, the exchange, the update; returns the summed loss (a 0-d
        device tensor of its own).  ``loss_view=True`` (single rank): return a VIEW of the loss slot instead -- no copy
        launch behind the step (5 us + a 9-us gap of the 1-ms one-view step) -- valid only until the next step()
        overwrites it: read it (``float()``) or clone it before stepping again"""
        if self.use_graph is None:
            nv = max(int(rot_local.shape[0]), 1)
            if self.d0.numel() * nv <= (2 << 20):    # small volumes, few views: the host needs longer than the GPU
                self.use_graph = True
            else:
                # Large volumes: ~80 launches per step.  At 8 views the GPU needs 2.7 ms for them and a fast host 2.0 ms to
                # issue them (tools/step_host_time.py: 35 us per C-ABI call, i.e. the HIP launches themselves) -- kernel-
                # bound, but a host of the same pool that is 2x slower (measured: 20 us per launch instead of 9) turns
                # the very same step host-bound, at 8 views as at 1.  So the choice is measured at the second step (the
                # first has built the lazy state); until then the step runs eagerly.
                self._steps_seen = getattr(self, "_steps_seen", 0) + 1
                if self._steps_seen == 2 and getattr(self, "d_s", None) is not None:
                    self.use_graph = self._loss_chain_host_bound(rot_local)
        if self.slab is not None:
            return self._step_slab(rot_local)
        self._apply_binding()
        if self.use_graph:
            total, g_ds = self._field_gradient_graphed(rot_local)
        elif self.pg is None:
            losses, g_ds = self.field_gradient(rot_local, total=True, for_variable=self.target == "v")
            total = self._loss_slot if losses is None else None
            total_new = None if losses is None else losses.sum()             # (one kernel, a fresh tensor: no slot, no copy)
        else:
            losses, g_ds = self.field_gradient(rot_local, total=True, for_variable=self.target == "v")
            if losses is not None:
                torch.sum(losses, dim=0, keepdim=True, out=self._loss_slot)  # (one kernel: reduce straight into the slot)
            total = self._loss_slot
        if self.pg is not None:
            # The one exchange step (RCCL over xGMI): ONE all-reduce of gradient + loss.  The reduction is placed on
            # the 4*G^3-byte density gradient, not on the 12*G^3-byte velocity gradient: everything below it is
            # linear and replicated, so reducing early moves 3x fewer bytes over the links.
            parallel.all_reduce_sum_([self._gbuf], group=self.pg)
        if total is not None:
            total = total[0] if (loss_view and self.pg is None) else total[0].clone()
        else:
            total = total_new
        if self._fused_step_ok():
            g_adv = ops.smooth3d_relu_bwd(self.d_s, g_ds, self.k)
            adv = self._adv_target()
            live = self._live_target()
            # (the mask buffer holds the CURRENT sample's mask when the adjoint above could use it: the Adam kernel may
            # then leave out the voxels that never were live)
            self.adam.step_through_advect(self.var, self.d0.unsqueeze(-1), g_adv.unsqueeze(-1), self.lr, adv_next=adv,
                                          live_next=live, live_current=live is not None and self._live_valid())
            if adv is not None:
                self._adv_mark(live=live is not None)
        else:
            self.adam.step(self.var, self.variable_gradient(g_ds), self.lr)
            self._adv_src = None          # (the variable moved and nothing wrote advect(d0, var) for it)
        return total


# --------------------------------------------------------------------------------------
# 2-D colour path (styler_2p.py:91-102 + styler_base.py:152-185, 211-213)
# --------------------------------------------------------------------------------------

class ImageStyleLoss(object):
    """Style (+TV) loss of a colour image d [B,H,W,3] in [0,1] (the 2-D colour stylizer):
    d*255 -> VGG -> Gram; with ``style_mask`` the features are multiplied by the bicubic-resized
    density mask and the Gram denominator becomes 2*area*C (styler_base.py:165-169)."""

    def __init__(self, net, style_layer, w_style_layer, w_style=1.0, w_tv=0.0, resize_scale=1.0,
                 style_mask=False, style_mask_on_ref=False, w_content=0.0, content_layer=None, content_channel=0,
                 w_content_amp=100.0, w_hist=0.0, hist_layer=(), w_hist_layer=()):
        assert not style_mask_on_ref, "style_mask_on_ref is not used by any reference driver"
        self.net = net
        self.layers = list(style_layer) if float(w_style) else []        # (styler_base.py:152: no style term at w_style 0)
        self.w_layers = [float(w) for w in w_style_layer][:len(self.layers)] if self.layers else []
        self.w_style, self.w_tv = float(w_style), float(w_tv)
        self.resize_scale = float(resize_scale)
        self.style_mask = bool(style_mask)
        order = [s[0] for s in net.seq]
        self.w_content = float(w_content) if content_layer else 0.0
        self.content_layer = content_layer if self.w_content else None
        self.content_channel = int(content_channel or 0)
        self.w_content_amp = float(w_content_amp)
        self.content_feature = None
        if self.content_layer is not None and dict((s[0], s[1]) for s in net.seq).get(self.content_layer) != "conv":
            raise KeyError("content_layer %r is not a conv layer of the loss network" % (self.content_layer,))
        self.v_batch = None                         # the content means run over the whole image batch (one sess.run)
        self.w_hist = float(w_hist)
        self.hist_layers = list(hist_layer) if self.w_hist else []
        self.w_hist_layers = [float(w) for w in w_hist_layer] if self.w_hist else []
        if len(self.w_hist_layers) == 1 and len(self.hist_layers) > 1:
            self.w_hist_layers = self.w_hist_layers * len(self.hist_layers)
        if len(self.w_hist_layers) != len(self.hist_layers):
            raise ValueError("w_hist_layer has %d entries for %d hist layers (one weight, or one per layer)"
                             % (len(self.w_hist_layers), len(self.hist_layers)))
        self.hist_targets = None
        vgg_hist = [n for n in self.hist_layers if "input" not in n]
        wanted = self.layers + vgg_hist + ([self.content_layer] if self.content_layer else [])
        if not wanted:
            raise ValueError("no loss term reaches the loss network: w_style 0 (or no style layer) and no content / "
                             "histogram layer")
        self.top = max(wanted, key=order.index)
        self.style_grams = None

    set_style_image = RenderStyleLoss.set_style_image
    set_content_image = RenderStyleLoss.set_content_image
    set_hist_image = RenderStyleLoss.set_hist_image
    _hist_job = RenderStyleLoss._hist_job
    _hist_input = RenderStyleLoss._hist_input
    _content_job = RenderStyleLoss._content_job
    _keep = RenderStyleLoss._keep
    _batch_views = RenderStyleLoss._batch_views
    out_hw = RenderStyleLoss.out_hw

    def d_img(self, d):
        B, H, W, _ = d.shape
        H2, W2 = self.out_hw(H, W)
        dimg, _ = ops.loss_net_input_fwd(d.contiguous(), H2, W2, want_x=False)
        return dimg

    def loss_and_grad(self, d, d_gray=None):
        """d [B,H,W,3]; d_gray [B,H,W,1] (mask, constant).  Returns (loss per image [B], dL/dd)."""
        B, H, W, _ = d.shape
        H2, W2 = self.out_hw(H, W)
        hist_in = any("input" in n for n in self.hist_layers)
        dimg, x = ops.loss_net_input_fwd(d.contiguous(), H2, W2, want_d_img=self.w_tv > 0 or hist_in)
        acts = self.net.forward(x, self.top, keep=self._keep())
        loss = torch.zeros(B, dtype=torch.float32, device=d.device)
        sg = {}
        for name, wl in zip(self.layers, self.w_layers):
            F = acts[name]
            _, h, w, cpad = F.shape
            c = _channels(acts, name, F)
            if self.style_mask:
                if not _post_relu(name):
                    raise NotImplementedError("style_mask on a '*_pre_relu' style layer (the mask adjoint folds the ReLU in)")
                m = ops.resize_bicubic_tf1(d_gray.contiguous(), h, w)      # [B,h,w,1], constant (no gradient to it)
                Fm, scale_dev = ops.style_mask_apply(F, m)                 # F * m, 1 / (2 area C)
                if c != cpad:
                    scale_dev = scale_dev * (float(cpad) / float(c))       # (rows padded with zero channels)
                G = ops.gram_fwd(Fm, 1.0, scale_dev=scale_dev)
                Dm = ops.style_loss_fwd(G, self.style_grams[name], wl * self.w_style, loss)
                dFm = ops.gram_bwd(Fm, Dm, 1.0, scale_dev=scale_dev, relu_mask=False)
                sg[name] = ops.style_mask_bwd(dFm, m, F)
            else:
                scale = 1.0 / (2.0 * h * w * c)
                G = ops.gram_fwd(F, scale)
                Dm = ops.style_loss_fwd(G, self.style_grams[name], wl * self.w_style, loss)
                sg[name] = ops.gram_bwd(F, Dm, scale, relu_mask=_post_relu(name))
        self._content_job(acts, sg, loss)
        hmask = d_gray if self.style_mask else None     # masked histogram branch (styler_base.py:196-201)
        self._hist_job(acts, sg, loss, hmask)
        g_x = self.net.backward(acts, sg, self.top)
        if hist_in:
            self._hist_input(dimg, loss, g_x, hmask)
        if self.w_tv > 0:
            tv = torch.zeros(1, dtype=torch.float32, device=d.device)
            ops.tv_loss(dimg, self.w_tv, tv, g_x)
            loss = loss + tv / B
        g_d = ops.loss_net_input_bwd(g_x, H, W, 3)
        return loss, g_d
