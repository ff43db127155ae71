"""Inception-v1 loss network -- the graph ``tensorflow_inception_graph.pb`` that the reference imports with
``tf.import_graph_def`` (styler_base.py:17-23, 51-57) and reads by tensor name (``_layer``, styler_base.py:91-94:
``graph.get_tensor_by_name("import/%s:0" % layer)``), assembled from the HIP node kernels (csrc/inception.hip) under
the graph's own node names.

The ``.pb`` is not shipped with the reference and TensorFlow is not in this image, so the graph cannot be read here.
What is restated is the PUBLISHED topology of that file (the "inception5h" / DeepDream Inception-v1: the node names
below are its node names; Szegedy et al. 2014, table 1, with the 5h file's own widths, e.g. mixed4a 3x3 = 204):

    input -> conv2d0 (7x7 / 2) -> maxpool0 (3x3 / 2) -> localresponsenorm0 -> conv2d1 (1x1) -> conv2d2 (3x3)
          -> localresponsenorm1 -> maxpool1 (3x3 / 2) -> mixed3a, mixed3b -> maxpool4 (3x3 / 2) -> mixed4a ... mixed4e
          -> maxpool10 (3x3 / 2) -> mixed5a, mixed5b
    mixedXY = concat(XY_1x1, XY_3x3(XY_3x3_bottleneck), XY_5x5(XY_5x5_bottleneck), XY_pool_reduce(XY_pool 3x3 / 1))

Every convolution unit ``U`` is the node triple ``U_pre_relu/conv`` (Conv2D, SAME) -> ``U_pre_relu`` (BiasAdd) -> ``U``
(Relu), with constants ``U_w`` [kh,kw,Cin,Cout] and ``U_b``.  Channel widths are NOT hard-wired: they are read from the
weight shapes (the table below only feeds the synthetic initialisation), and the LRN attributes come from the weight
file when it carries them.  Addressable tensors: ``U``, ``U_pre_relu``, the pools, the LRNs and the module outputs.
The main classifier (``avgpool0`` -> ``softmax2_pre_activation``, the logits the reference's ``top_k`` content target
reads) is built; the two auxiliary heads (head0 / head1, nn*, softmax0 / softmax1) are on the path to no tensor the
reference asks for and are not.

Weights: ``tensorflow_inception_graph.npz`` next to the ``.pb`` path, converted offline with keys
``<unit>_w`` / ``<unit>_b`` (the names of the graph's Const nodes) and optionally ``localresponsenorm0`` /
``localresponsenorm1`` = [depth_radius, bias, alpha, beta].  Without the file the loader RAISES unless seeded synthetic
weights are asked for explicitly (config.synthetic_weights / NFS_SYNTHETIC_VGG=1), as for VGG.

Layout: NHWC float32; a module output lives in ONE buffer whose rows are padded to a multiple of 64 floats (zeros
behind the logical channels: the Gram kernels take multiples of 64), the branches write their channel ranges in
place, data gradients read them in place.  ``acts[name]`` is always such a padded contiguous tensor;
``acts.channels[name]`` is the logical channel count.
"""
from __future__ import annotations

import math
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

from . import ops

# (1x1, 3x3_bottleneck, 3x3, 5x5_bottleneck, 5x5, pool_reduce) of the published 5h graph -- synthetic initialisation only
MIXED_WIDTHS = OrderedDict([
    ("mixed3a", (64, 96, 128, 16, 32, 32)), ("mixed3b", (128, 128, 192, 32, 96, 64)),
    ("mixed4a", (192, 96, 204, 16, 48, 64)), ("mixed4b", (160, 112, 224, 24, 64, 64)),
    ("mixed4c", (128, 128, 256, 24, 64, 64)), ("mixed4d", (112, 144, 288, 32, 64, 64)),
    ("mixed4e", (256, 160, 320, 32, 128, 128)), ("mixed5a", (256, 160, 320, 48, 128, 128)),
    ("mixed5b", (384, 192, 384, 48, 128, 128))])
STEM_WIDTHS = OrderedDict([("conv2d0", (7, 64)), ("conv2d1", (1, 64)), ("conv2d2", (3, 192))])
# tf.nn.lrn attributes of the two LRN nodes of the 5h graph: (depth_radius, bias, alpha, beta); a weight file that
# carries them overrides these
LRN_DEFAULT = (5, 2.0, 1e-4, 0.5)

# units in graph order: (kind, name[, stride])
UNITS = ([("conv", "conv2d0"), ("maxpool", "maxpool0", 2), ("lrn", "localresponsenorm0"), ("conv", "conv2d1"),
          ("conv", "conv2d2"), ("lrn", "localresponsenorm1"), ("maxpool", "maxpool1", 2), ("mixed", "mixed3a"),
          ("mixed", "mixed3b"), ("maxpool", "maxpool4", 2)] + [("mixed", "mixed4" + c) for c in "abcde"]
         + [("maxpool", "maxpool10", 2), ("mixed", "mixed5a"), ("mixed", "mixed5b"),
            # the main classifier: AvgPool 7x7 VALID, reshape to [-1, 1024], MatMul + BiasAdd -> the logits the reference's
            # top_k content target reads (styler_base.py:240-245); needs >= 193 input pixels a side (7 x 7 at mixed5b)
            ("avgpool", "avgpool0", 7), ("fc", "softmax2_pre_activation")])
BRANCHES = ("1x1", "3x3_bottleneck", "3x3", "5x5_bottleneck", "5x5", "pool_reduce")
# modules of up to this many pixels (batch x h x w) run their branches as grouped launches (64-row tiles, K split until
# the launch fills the chip); above it every branch is a full launch of its own
GROUP_MAX_PIXELS = 1 << 16


def conv_units():
    """names of the convolution units in graph order"""
    out = []
    for u in UNITS:
        if u[0] == "conv":
            out.append(u[1])
        elif u[0] == "mixed":
            out.extend("%s_%s" % (u[1], b) for b in BRANCHES)
    return out


def synthetic_weights(seed=123, upto=None):
    """w ~ N(0, 2 / (kh kw Cin)) HWIO, b ~ 0.01 N(0,1), drawn unit by unit from RandomState(seed)"""
    rng = np.random.RandomState(seed)
    out = OrderedDict()

    def draw(name, k, cin, cout):
        w = rng.randn(k, k, cin, cout) * math.sqrt(2.0 / (k * k * cin))
        b = rng.randn(cout) * 0.01
        out[name] = (w.astype(np.float32), b.astype(np.float32))

    cin = 3
    for u in UNITS:
        if u[0] == "conv":
            k, cout = STEM_WIDTHS[u[1]]
            draw(u[1], k, cin, cout)
            cin = cout
        elif u[0] == "mixed":
            c1, c3b, c3, c5b, c5, cp = MIXED_WIDTHS[u[1]]
            draw(u[1] + "_1x1", 1, cin, c1)
            draw(u[1] + "_3x3_bottleneck", 1, cin, c3b)
            draw(u[1] + "_3x3", 3, c3b, c3)
            draw(u[1] + "_5x5_bottleneck", 1, cin, c5b)
            draw(u[1] + "_5x5", 5, c5b, c5)
            draw(u[1] + "_pool_reduce", 1, cin, cp)
            cin = c1 + c3 + c5 + cp
        elif u[0] == "fc":
            draw("softmax2", 1, cin, 1008)
        if u[1] == upto:
            break
    return out


def unit_of(name):
    """the graph unit (entry of UNITS) that produces tensor ``name``"""
    base = name[:-len("_pre_relu")] if name.endswith("_pre_relu") else name
    for u in UNITS:
        if base == u[1] or (u[0] == "mixed" and base.startswith(u[1] + "_")):
            return u[1]
    if name == "avgpool0/reshape":
        return "avgpool0"
    raise KeyError("%r is not a tensor of the Inception-v1 graph" % (name,))


def load_npz_weights(path, upto=None):
    """(weights, lrn attributes) from the offline conversion of the graph's Const nodes; every convolution unit down
    to ``upto`` must be present, 4-D HWIO with a bias of matching length"""
    z = np.load(path)
    out = OrderedDict()
    last = unit_of(upto) if upto else None
    stop = False
    for u in UNITS:
        names = [u[1]] if u[0] == "conv" else ["%s_%s" % (u[1], b) for b in BRANCHES] if u[0] == "mixed" else []
        if u[0] == "fc" and "softmax2_w" in z and "softmax2_b" in z:      # (the classifier is optional in a weight file)
            w2 = np.asarray(z["softmax2_w"], np.float32)
            out["softmax2"] = (w2.reshape((1, 1) + w2.shape[-2:]), np.asarray(z["softmax2_b"], np.float32).reshape(-1))
        for name in names:
            wk, bk = name + "_w", name + "_b"
            if wk not in z or bk not in z:
                raise KeyError("%s: no weights for unit %s (keys %s, %s)" % (path, name, wk, bk))
            w, b = np.asarray(z[wk], np.float32), np.asarray(z[bk], np.float32).reshape(-1)
            if w.ndim != 4 or w.shape[0] != w.shape[1] or b.shape != (w.shape[3],):
                raise ValueError("%s: unit %s has shapes %s / %s, expected [k,k,Cin,Cout] HWIO / [Cout]"
                                 % (path, name, w.shape, b.shape))
            out[name] = (w, b)
        if u[1] == last:
            stop = True
        if stop:
            break
    lrn = {}
    for name in ("localresponsenorm0", "localresponsenorm1"):
        if name in z:
            a = np.asarray(z[name], np.float64).reshape(-1)
            if a.shape != (4,):
                raise ValueError("%s: %s must hold [depth_radius, bias, alpha, beta]" % (path, name))
            lrn[name] = (int(a[0]), float(a[1]), float(a[2]), float(a[3]))
    return out, lrn


def _pad64(c):
    return (c + 63) // 64 * 64


class _Activations(dict):
    """name -> padded contiguous activation [B,h,w,ld]; ``channels[name]`` = logical channel count.  Internals of the
    pass (pool arguments, LRN scales, branch buffers) ride along for the data gradient."""

    def __init__(self):
        dict.__init__(self)
        self.channels = {}
        self.where = {}          # tensor name -> (buffer, first channel, channels) where it really lives
        self.aux = {}
        self.relu_bits = {}
        self.hw = {}


class InceptionV1(object):
    """forward keeps what the data gradient needs; backward is data-gradient only (frozen weights)"""

    def __init__(self, weights, device, lrn=None, pool1=False):
        self.device = torch.device(device)
        # styler_base.py:25-30: ``pool1`` sets the first convolution's stride to 1 ("fix checkerboard artifacts")
        self.stride0 = 1 if pool1 else 2
        self.lrn = {"localresponsenorm0": LRN_DEFAULT, "localresponsenorm1": LRN_DEFAULT}
        self.lrn.update(lrn or {})
        self.params = {}
        cin = 3
        self.cout = {}                                   # tensor name -> logical channels
        self.seq = []
        have = set(weights.keys())

        def unit(name, cin_expected, ksizes):
            if name not in have:
                return None
            w, b = weights[name]
            k = w.shape[0]
            if k not in ksizes or w.shape[2] != cin_expected:
                raise ValueError("unit %s: filter %s does not fit the graph here (kernel %s, %d input channels)"
                                 % (name, tuple(w.shape), "/".join(str(s) for s in ksizes), cin_expected))
            if w.shape[3] % 4:
                raise ValueError("unit %s: %d output channels (a multiple of 4 is needed)" % (name, w.shape[3]))
            wt = torch.as_tensor(w, dtype=torch.float32).to(self.device).contiguous()
            p = dict(k=k, cin=w.shape[2], cout=w.shape[3], w=wt, fwd=ops.conv2d_pack(wt, False),
                     bias=torch.as_tensor(b, dtype=torch.float32).to(self.device).contiguous())
            if p["cin"] > 4:
                p["dgrad"] = ops.conv2d_pack(wt, True)
            self.params[name] = p
            self.cout[name] = self.cout[name + "_pre_relu"] = p["cout"]
            self.seq.append((name + "_pre_relu", "conv", p["cin"], p["cout"]))
            self.seq.append((name, "conv", p["cin"], p["cout"]))
            return p

        self.units = []
        for u in UNITS:
            if u[0] == "conv":
                p = unit(u[1], cin, (7,) if u[1] == "conv2d0" else (1, 3, 5))
                if p is None:
                    break
                cin = p["cout"]
            elif u[0] == "mixed":
                ps = [unit("%s_%s" % (u[1], b), cin if b in ("1x1", "3x3_bottleneck", "5x5_bottleneck", "pool_reduce")
                           else self.params["%s_%s_bottleneck" % (u[1], b)]["cout"], (1, 3, 5))
                      if ("%s_%s" % (u[1], b)) in have else None for b in BRANCHES]
                if any(p is None for p in ps):
                    break
                self.cout[u[1] + "_pool"] = cin
                self.seq.insert(len(self.seq) - 2, (u[1] + "_pool", "pool", cin, cin))
                cin = ps[0]["cout"] + ps[2]["cout"] + ps[4]["cout"] + ps[5]["cout"]
                self.cout[u[1]] = cin
                self.seq.append((u[1], "conv", cin, cin))
            elif u[0] == "fc":
                if "softmax2" not in have:
                    break
                w2, b2 = weights["softmax2"]
                if w2.shape[:3] != (1, 1, cin) or w2.shape[3] % 4:
                    raise ValueError("softmax2: weights %s do not fit the %d pooled channels" % (tuple(w2.shape), cin))
                wt = torch.as_tensor(w2, dtype=torch.float32).to(self.device).contiguous()
                self.params["softmax2"] = dict(k=1, cin=cin, cout=w2.shape[3], w=wt, fwd=ops.conv2d_pack(wt, False),
                                               dgrad=ops.conv2d_pack(wt, True),
                                               bias=torch.as_tensor(b2, dtype=torch.float32).to(self.device).contiguous())
                cin = w2.shape[3]
                self.cout[u[1]] = cin
                self.seq.append((u[1], "conv", self.params["softmax2"]["cin"], cin))
            else:
                self.cout[u[1]] = cin
                self.seq.append((u[1], "pool", cin, cin))
            self.units.append(u)
        if not self.units:
            raise ValueError("no weights for conv2d0: not an Inception-v1 weight set")
        self.names = [s[0] for s in self.seq]

    # -- what the engine asks of a loss network ---------------------------------------------------------------------
    def masks_addend_of(self, name, acts_shape):
        """(VGG: a style gradient may be handed over without its ReLU mask.)  Here every ReLU adjoint is applied where
        the gradient is consumed, whatever was injected: masking twice is idempotent, so the caller may mask."""
        return False

    def _last_unit(self, upto):
        u = unit_of(upto)
        idx = [i for i, x in enumerate(self.units) if x[1] == u]
        if not idx:
            raise KeyError("no weights down to %s" % upto)
        return idx[0]

    def forward(self, x, upto, on_layer=None, keep=None):
        """x [B,H,W,3] (mean-subtracted, vgg.preprocess -- styler_base.py:54 feeds the same input to both networks)
        -> activations of every unit down to the one that produces ``upto``.  ``keep``: the tensor names the caller
        will read (``*_pre_relu`` tensors and branch tensors are only materialised when named; None = all of them)"""
        if upto not in self.cout:
            raise KeyError(upto)
        acts = _Activations()
        B = x.shape[0]
        want_pre = (lambda n: True) if keep is None else (lambda n: (n + "_pre_relu") in keep or n + "_pre_relu" == upto)
        cur, ccur = x.contiguous(), 3                     # current tensor (padded rows) and its logical channels
        acts.aux["input_hw"] = (x.shape[1], x.shape[2])

        def conv(name, src, csrc, dst, cdst, stride=1, group=None):
            """one convolution unit; with ``group`` (a list) the call is queued for a grouped launch"""
            p = self.params[name]
            pre = None
            if want_pre(name):
                pre = torch.empty(tuple(dst.shape[:3]) + (_pad64(p["cout"]),), dtype=torch.float32, device=x.device)
                if pre.shape[3] > p["cout"]:
                    pre.zero_()
                acts[name + "_pre_relu"] = pre
                acts.channels[name + "_pre_relu"] = p["cout"]
                acts.where[name + "_pre_relu"] = (pre, 0, p["cout"])
            if group is not None:
                group.append(dict(x=src, cx=csrc, Cin=p["cin"], packed=p["fwd"], bias=p["bias"], y=dst, cy=cdst,
                                  Cout=p["cout"], k=p["k"], relu=True, y_pre=pre))
            else:
                ops.conv2d_fwd(src, csrc, p["cin"], p["fwd"], p["bias"], dst, cdst, p["cout"], p["k"], p["k"], stride,
                               relu=True, y_pre=pre)
            acts.where[name] = (dst, cdst, p["cout"])

        def new(b, h, w, c):
            ld = _pad64(c)
            t = torch.empty((b, h, w, ld), dtype=torch.float32, device=x.device)
            if ld > c:
                t.zero_()
            return t

        last = self._last_unit(upto)
        for u in self.units[:last + 1]:
            kind, name = u[0], u[1]
            H, W = cur.shape[1], cur.shape[2]
            if kind == "conv":
                p = self.params[name]
                s = self.stride0 if name == "conv2d0" else 1
                out = new(B, ops.same_out(H, p["k"], s)[0], ops.same_out(W, p["k"], s)[0], p["cout"])
                conv(name, cur, 0, out, 0, s)
                acts.aux[name + "/in"] = (cur, ccur)
                cur, ccur = out, p["cout"]
            elif kind == "maxpool":
                out, arg = ops.maxpool3_fwd(cur, u[2])
                acts.aux[name] = (arg, (H, W), u[2])
                acts.aux[name + "/in"] = (cur, ccur)
                cur = out
                acts.where[name] = (out, 0, ccur)
            elif kind == "avgpool":
                if H < u[2] or W < u[2]:
                    raise ValueError("%s: the %d x %d map in front of the classifier is smaller than its %d x %d window "
                                     "(the graph classifies 224 x 224 images)" % (name, H, W, u[2], u[2]))
                out = ops.avgpool_valid_fwd(cur, u[2])
                acts.aux[name] = ((H, W), u[2])
                acts.aux[name + "/in"] = (cur, ccur)
                cur = out
                acts.where[name] = (out, 0, ccur)
            elif kind == "fc":
                p = self.params["softmax2"]
                out = new(B, H, W, p["cout"])
                ops.conv2d_fwd(cur, 0, p["cin"], p["fwd"], p["bias"], out, 0, p["cout"], 1, 1, 1, relu=False)
                acts.aux[name + "/in"] = (cur, ccur)
                cur, ccur = out, p["cout"]
                acts.where[name] = (out, 0, ccur)
            elif kind == "lrn":
                r, bias, alpha, beta = self.lrn[name]
                out, scale = ops.lrn_fwd(cur, ccur, r, bias, alpha, beta)
                acts.aux[name] = (cur, out, scale)
                acts.aux[name + "/in"] = (cur, ccur)
                cur = out
                acts.where[name] = (out, 0, ccur)
            else:
                ps = [self.params["%s_%s" % (name, b)] for b in BRANCHES]
                c1, c3, c5, cp = ps[0]["cout"], ps[2]["cout"], ps[4]["cout"], ps[5]["cout"]
                out = new(B, H, W, c1 + c3 + c5 + cp)
                # (the bottleneck outputs are read by their own branch only: exact width, no padding to fill)
                b3 = torch.empty((B, H, W, ps[1]["cout"]), dtype=torch.float32, device=x.device)
                b5 = torch.empty((B, H, W, ps[3]["cout"]), dtype=torch.float32, device=x.device)
                # a module = two grouped launches (the three 1x1 convolutions on the module input; 3x3, 5x5 and the
                # pool projection) + the pool: one by one the branches leave most of the chip idle
                grouped = B * H * W <= GROUP_MAX_PIXELS
                g1 = [] if grouped else None
                conv(name + "_1x1", cur, 0, out, 0, group=g1)
                conv(name + "_3x3_bottleneck", cur, 0, b3, 0, group=g1)
                conv(name + "_5x5_bottleneck", cur, 0, b5, 0, group=g1)
                if grouped:
                    ops.conv2d_group(g1)
                pool, arg = ops.maxpool3_fwd(cur, 1)
                acts.aux[name + "_pool"] = (arg, (H, W), 1)
                acts.where[name + "_pool"] = (pool, 0, ccur)
                g2 = [] if grouped else None
                conv(name + "_3x3", b3, 0, out, c1, group=g2)
                conv(name + "_5x5", b5, 0, out, c1 + c3, group=g2)
                conv(name + "_pool_reduce", pool, 0, out, c1 + c3 + c5, group=g2)
                if grouped:
                    ops.conv2d_group(g2)
                acts.aux[name + "/in"] = (cur, ccur)
                cur, ccur = out, c1 + c3 + c5 + cp
                acts.where[name] = (out, 0, ccur)
            if kind != "mixed" and kind != "conv":
                pass
            # the tensors a caller may read by name: whole buffers as they are, channel ranges as padded copies
            for n, (buf, c0, c) in list(acts.where.items()):
                if n in acts:
                    continue
                if c0 == 0 and buf.shape[3] == _pad64(c):
                    acts[n] = buf
                elif keep is None or n in keep or n == upto:
                    t = new(B, buf.shape[1], buf.shape[2], c)
                    t[..., :c] = buf[..., c0:c0 + c]
                    acts[n] = t
                acts.channels[n] = c
            if on_layer is not None and name in acts:
                on_layer(name, acts[name])
        return acts

    def backward(self, acts, grads, upto, unmasked=()):
        """grads: tensor name -> dL/d(tensor) in the layout of ``acts[name]`` (padded rows; whether a gradient at a
        post-ReLU tensor already carries that ReLU's mask makes no difference).  Returns dL/dx [B,H,W,3]."""
        G = {}                                              # id(buffer) -> gradient buffer of the same shape
        pre_inject = {}
        premasked = set()                                   # gradient buffers that already carry their ReLU adjoint

        def gbuf(buf):
            g = G.get(id(buf))
            if g is None:
                g = G[id(buf)] = torch.zeros_like(buf)
            return g

        def target(buf):
            """(gradient buffer of ``buf``, accumulate?): the first term of a gradient is WRITTEN into a fresh buffer (no
            zero fill: 25 fills per pass otherwise), later terms are added.  Channels behind the logical width of a
            padded buffer are never written and never read as values."""
            g = G.get(id(buf))
            if g is None:
                G[id(buf)] = g = torch.empty_like(buf)
                return g, False
            return g, True

        for n, g in grads.items():
            if g is None:
                continue
            if n.endswith("_pre_relu"):
                pre_inject[n[:-len("_pre_relu")]] = g
                continue
            buf, c0, c = acts.where[n]
            if c0 == 0 and tuple(g.shape) == tuple(buf.shape) and id(buf) not in G:
                G[id(buf)] = g                              # the caller's tensor IS the gradient buffer (it is not kept)
                continue
            gb = gbuf(buf)
            if c0 == 0 and tuple(g.shape) == tuple(gb.shape):
                gb.add_(g)
            else:
                gb[..., c0:c0 + c].add_(g[..., :c])

        def conv_bwd(name, src, csrc, group=None):
            """data gradient of unit ``name`` into the gradient of its input (src buffer, logical channels csrc); with
            ``group`` (a list) the call is queued for a grouped launch"""
            p = self.params[name]
            buf, c0, c = acts.where[name]
            g = G.get(id(buf))
            inj = pre_inject.get(name)
            if g is None and inj is None:
                return
            if inj is not None:
                g_in, cg, mask, cm = ops.relu_mask_add(g, c0, buf if g is not None else None, c0, inj, 0, c), 0, None, 0
            else:
                g_in, cg, mask, cm = g, c0, buf, c0
            dst, acc = target(src)
            if group is not None:
                group.append(dict(x=g_in, cx=cg, Cin=c, packed=p["dgrad"], bias=None, y=dst, cy=0, Cout=csrc,
                                  k=p["k"], relu=False, x_mask=mask, cm=cm, accumulate=acc))
            else:
                ops.conv2d_fwd(g_in, cg, c, p["dgrad"], None, dst, 0, csrc, p["k"], p["k"], 1, relu=False,
                               x_mask=mask, cm=cm, accumulate=acc)

        last = self._last_unit(upto)
        for u in reversed(self.units[:last + 1]):
            kind, name = u[0], u[1]
            src, csrc = acts.aux[name + "/in"]
            if kind == "conv" and name == "conv2d0":
                p = self.params[name]
                buf, c0, c = acts.where[name]
                g = G.get(id(buf))
                inj = pre_inject.get(name)
                if g is None and inj is None:
                    return torch.zeros(tuple(src.shape[:3]) + (3,), dtype=torch.float32, device=src.device)
                if inj is not None:
                    g, mask = ops.relu_mask_add(g, 0, buf if (g is not None and id(buf) not in premasked) else None, 0,
                                                inj, 0, c), None
                else:
                    mask = None if id(buf) in premasked else buf
                return ops.conv2d_dgrad_small(g, 0, c, p["w"], acts.aux["input_hw"], self.stride0, y_act=mask)
            if kind == "conv":
                conv_bwd(name, src, csrc)
            elif kind == "maxpool":
                arg, hw, stride = acts.aux[name]
                g = G.get(id(acts.where[name][0]))
                if g is not None:
                    # the pool over conv2d0's output is the last term of that gradient: it hands it on with the ReLU
                    # adjoint, and the data gradient down to the image (16 taps per pixel) reads it unmasked
                    first = name == "maxpool0"
                    r = ops.maxpool3_bwd(g, arg, hw, stride, gx=G.get(id(src)), relu_of=src if first else None)
                    G[id(src)] = r
                    if first:
                        premasked.add(id(src))
            elif kind == "avgpool":
                hw, k = acts.aux[name]
                g = G.get(id(acts.where[name][0]))
                if g is not None:
                    G[id(src)] = ops.avgpool_valid_bwd(g, hw, k, gx=G.get(id(src)))
            elif kind == "fc":
                p = self.params["softmax2"]
                g = G.get(id(acts.where[name][0]))
                if g is not None:
                    dst, acc = target(src)
                    ops.conv2d_fwd(g, 0, p["cout"], p["dgrad"], None, dst, 0, p["cin"], 1, 1, 1, relu=False,
                                   accumulate=acc)
            elif kind == "lrn":
                xin, y, scale = acts.aux[name]
                g = G.get(id(y))
                if g is not None:
                    r, bias, alpha, beta = self.lrn[name]
                    G[id(src)] = ops.lrn_bwd(xin, y, scale, g, csrc, r, alpha, beta, gx=G.get(id(src)))
            else:
                b3 = acts.where[name + "_3x3_bottleneck"][0]
                b5 = acts.where[name + "_5x5_bottleneck"][0]
                pool = acts.where[name + "_pool"][0]
                grouped = src.shape[0] * src.shape[1] * src.shape[2] <= GROUP_MAX_PIXELS
                # the three data gradients that end in other tensors in one launch; then the three 1x1 data gradients
                # that meet in the gradient of the module input as ONE sum (partial sums + a fixed-order reduction: no
                # race between them, deterministic)
                g2 = [] if grouped else None
                conv_bwd(name + "_3x3", b3, self.params[name + "_3x3_bottleneck"]["cout"], group=g2)
                conv_bwd(name + "_5x5", b5, self.params[name + "_5x5_bottleneck"]["cout"], group=g2)
                conv_bwd(name + "_pool_reduce", pool, csrc, group=g2)
                if g2:
                    ops.conv2d_group(g2)
                g1 = [] if grouped else None
                conv_bwd(name + "_1x1", src, csrc, group=g1)
                conv_bwd(name + "_3x3_bottleneck", src, csrc, group=g1)
                conv_bwd(name + "_5x5_bottleneck", src, csrc, group=g1)
                if g1:
                    for q in g1[1:]:
                        q["sum_with_prev"] = True
                    ops.conv2d_group(g1)
                gp = G.get(id(pool))
                if gp is not None:
                    arg, hw, stride = acts.aux[name + "_pool"]
                    G[id(src)] = ops.maxpool3_bwd(gp, arg, hw, stride, gx=G.get(id(src)))
        raise AssertionError("unreachable")


def load_inception(model_path, device, seed=123, synthetic=None, pool1=False):
    """Counterpart of styler_base.py:17-30 (GraphDef read + the ``pool1`` stride edit).  ``model_path``
    '.../tensorflow_inception_graph.pb' -> the Const nodes converted offline to '.../tensorflow_inception_graph.npz'
    (a GraphDef cannot be parsed without TensorFlow / protobuf definitions here).  Without that file the loader
    RAISES unless seeded synthetic weights are asked for explicitly; ``net.source`` names what was loaded."""
    npz = os.path.splitext(model_path)[0] + ".npz"
    if synthetic is None:
        synthetic = os.environ.get("NFS_SYNTHETIC_VGG", "0") == "1"
    lrn = None
    if os.path.exists(npz):
        w, lrn = load_npz_weights(npz)
        src = npz
    elif synthetic:
        w = synthetic_weights(seed)
        src = "synthetic(seed=%d)" % seed
    else:
        raise FileNotFoundError(
            "%s not found%s.  Convert the graph's Const nodes to an .npz with keys '<unit>_w' [k,k,Cin,Cout], "
            "'<unit>_b' (conv2d0 ... mixed5b_pool_reduce; INTEGRATION.md), or opt in to seeded SYNTHETIC weights "
            "with config.synthetic_weights=True / NFS_SYNTHETIC_VGG=1 (results are then not a stylisation by "
            "Inception-v1)" % (npz, " (the GraphDef %s is present but cannot be parsed without TensorFlow)" % model_path
                               if os.path.exists(model_path) else ""))
    net = InceptionV1(w, device, lrn=lrn, pool1=pool1)
    net.source = src
    print("loss network: Inception-v1 weights from %s" % src, file=sys.stderr)
    return net
