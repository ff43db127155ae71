"""VGG-19 loss network (slim variant with average pooling) -- host-side mirror of the
reference's ``vgg.py`` (vgg.py:44-120) driving the HIP conv / pool kernels.

Weights are frozen.  They come from (a) an ``.npz`` converted offline from
``vgg_19_2016_08_28`` (keys ``conv{b}_{i}/weights`` HWIO and ``conv{b}_{i}/biases``) or
(b) a seeded synthetic He-normal initialisation (the checkpoint is not shipped with the
reference and there is no network here).
"""
from __future__ import annotations

import math
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

from . import ops

# vgg.py:18-20 -- mean only (std is commented out in the reference), RGB order
_R_MEAN = 0.485 * 255
_G_MEAN = 0.456 * 255
_B_MEAN = 0.406 * 255

VGG19_BLOCKS = (("conv1", 2, 64), ("conv2", 2, 128), ("conv3", 4, 256), ("conv4", 4, 512), ("conv5", 4, 512))
VGG16_BLOCKS = (("conv1", 2, 64), ("conv2", 2, 128), ("conv3", 3, 256), ("conv4", 3, 512), ("conv5", 3, 512))


def layer_sequence(blocks=VGG19_BLOCKS):
    """[('conv1_1', 'conv', 3, 64), ..., ('pool1', 'pool', 64, 64), ...] in network order"""
    seq, cin = [], 3
    for blk, reps, cout in blocks:
        for i in range(reps):
            seq.append(("%s_%d" % (blk, i + 1), "conv", cin, cout))
            cin = cout
        seq.append(("pool" + blk[-1], "pool", cout, cout))
    return seq


def synthetic_weights(seed=123, blocks=VGG19_BLOCKS, upto=None):
    """w ~ N(0, 2/(9 Cin)) HWIO, b ~ 0.01 N(0,1), drawn layer by layer from RandomState(seed)."""
    rng = np.random.RandomState(seed)
    out = OrderedDict()
    for name, kind, cin, cout in layer_sequence(blocks):
        if kind != "conv":
            continue
        w = rng.randn(3, 3, cin, cout) * math.sqrt(2.0 / (9 * cin))
        b = rng.randn(cout) * 0.01
        out[name] = (w.astype(np.float32), b.astype(np.float32))
        if name == upto:
            break
    return out


def load_npz_weights(path, blocks=VGG19_BLOCKS, scope="vgg_19", upto=None):
    """Weights converted offline from the slim checkpoint (``vgg_19_2016_08_28``).  Accepted key forms per layer:
    ``conv1_1/weights`` + ``conv1_1/biases``, the checkpoint's own ``<scope>/conv1/conv1_1/weights`` (+ ``biases``; scope
    = model name, vgg.py:89,119) or ``conv1_1_w`` + ``conv1_1_b``.  Every conv layer of the network (down to ``upto``
    if given) must be present with shape [3,3,Cin,Cout] HWIO: a missing layer raises instead of silently truncating
    the network."""
    z = np.load(path)
    out = OrderedDict()
    for name, kind, cin, cout in layer_sequence(blocks):
        if kind != "conv":
            continue
        forms = (("%s/weights" % name, "%s/biases" % name),
                 ("%s/%s/%s/weights" % (scope, name[:5], name), "%s/%s/%s/biases" % (scope, name[:5], name)),
                 (name + "_w", name + "_b"))
        for wk, bk in forms:
            if wk in z and bk in z:
                break
        else:
            raise KeyError("%s: no weights for layer %s (looked for %s)" % (path, name, ", ".join(f[0] for f in forms)))
        w, b = np.asarray(z[wk], np.float32), np.asarray(z[bk], np.float32)
        if w.shape != (3, 3, cin, cout) or b.shape != (cout,):
            raise ValueError("%s: layer %s has shapes %s / %s, expected (3,3,%d,%d) HWIO / (%d,)"
                             % (path, name, w.shape, b.shape, cin, cout, cout))
        out[name] = (w, b)
        if name == upto:
            break
    return out


class _Activations(OrderedDict):
    """name -> activation of one forward pass, plus that pass's per-layer ReLU bit caches"""

    def __init__(self):
        OrderedDict.__init__(self)
        self.relu_bits = {}
        self.hw = {}                    # (H, W) of pooled layers whose full-resolution output was not materialised


class VGG(object):
    """Forward caches the post-ReLU activations (end points, vgg.py:55-66); backward is
    data-gradient only (weights frozen)."""

    def __init__(self, weights, device, blocks=VGG19_BLOCKS):
        self.device = torch.device(device)
        self.seq = [s for s in layer_sequence(blocks)]
        self.params = {}
        for name, (w, b) in weights.items():
            wt = torch.as_tensor(w, dtype=torch.float32).to(self.device).contiguous()
            self.params[name] = dict(
                fwd=ops.conv3x3_pack(wt, 0), dgrad=ops.conv3x3_pack(wt, 1),
                bias=torch.as_tensor(b, dtype=torch.float32).to(self.device).contiguous(),
                cin=wt.shape[2], cout=wt.shape[3])
        self.names = list(weights.keys())

    def plan(self, upto):
        idx = [i for i, s in enumerate(self.seq) if s[0] == upto]
        if not idx:
            raise KeyError(upto)
        plan = self.seq[:idx[0] + 1]
        for name, kind, _, _ in plan:
            if kind == "conv" and name not in self.params:
                raise KeyError("no weights for %s" % name)
        return plan

    def forward(self, x, upto, on_layer=None, keep=None):
        """x [B,H,W,3] (mean-subtracted) -> OrderedDict name -> [B,h,w,C] post-ReLU / pooled.
        ``on_layer(name, tensor)`` is called right after a layer has been enqueued (the style loss uses it to
        start that layer's Gram work on a second stream while the next convolutions run)."""
        acts = _Activations()
        cur = x
        plan = self.plan(upto)
        pooled = None
        bits = acts.relu_bits           # name -> the layer's ReLU bit cache (masks for its data gradient), if it keeps one
        for li, (name, kind, cin, cout) in enumerate(plan):
            if kind == "conv":
                p = self.params[name]
                nxt_pool = li + 1 < len(plan) and plan[li + 1][1] == "pool"
                B_, H_, W_ = cur.shape[0], cur.shape[1], cur.shape[2]
                below_conv = li > 0 and plan[li - 1][1] == "conv"          # the data gradient masks with (x > 0)
                if nxt_pool and cin % 32 == 0 and cur.shape[1] >= 2 and cur.shape[2] >= 2:
                    # the conv feeds a 2x2 average pool: both outputs from one pass (pool folded into the transform)
                    rb = ops.conv3x3_relu_bits(B_, H_, W_, cin, cout, True, cur.device) if below_conv else None
                    # nothing but the pool and this layer's own data gradient (served by the bit cache) reads the
                    # full-resolution output of a pooled layer, unless the caller wants it (``keep``)
                    want_y = rb is None or keep is None or name in keep
                    acts.hw[name] = (H_, W_)
                    cur, pooled = ops.conv3x3_fwd_pool(cur, p["fwd"], p["bias"], cout, relu=True, relu_bits=rb,
                                                       want_y=want_y)
                else:
                    rb = ops.conv3x3_relu_bits(B_, H_, W_, cin, cout, False, cur.device) if below_conv else None
                    cur = ops.conv3x3_fwd(cur, p["fwd"], p["bias"], cout, relu=True, relu_bits=rb)
                if rb is not None:
                    bits[name] = rb
            else:
                cur = pooled if pooled is not None else ops.avgpool2_fwd(cur)
                pooled = None
            acts[name] = cur
            if on_layer is not None:
                on_layer(name, cur)
        return acts

    def masks_addend_of(self, name, acts_shape):
        """True when the gradient injected at conv layer ``name`` (a style / content gradient) may be handed over
        WITHOUT its ReLU mask: the data gradient of the conv above adds it before applying that same mask (Winograd
        F(4x4) path with the ReLU bit cache).  ``acts_shape`` = [B,h,w,C] of the layer's activation."""
        names = [s_[0] for s_ in self.seq]
        i = names.index(name)
        if i + 1 >= len(self.seq) or self.seq[i + 1][1] != "conv":
            return False
        _, _, cin, cout = self.seq[i + 1]
        B, h, w, _ = acts_shape
        nxt_pool = i + 2 < len(self.seq) and self.seq[i + 2][1] == "pool"
        return ops.conv3x3_relu_bits_words(B, h, w, cin, cout, nxt_pool) > 0

    def backward(self, acts, style_grads, upto, unmasked=()):
        """style_grads: name -> dL/d(pre-activation contribution) already masked by (act > 0)
        (ops.gram_bwd(relu_mask=True)), except the names in ``unmasked`` (see masks_addend_of), whose mask is applied
        by the data gradient that adds them.  Returns dL/dx [B,H,W,3]."""
        assert upto not in unmasked, "the top layer's gradient enters the chain directly: it must be masked"
        plan = self.plan(upto)
        g = style_grads[upto]           # gradient wrt the pre-activation of the top conv
        pooled_from = None              # set when g is still at the pooled resolution below conv `pooled_from`
        for li in range(len(plan) - 1, -1, -1):
            name, kind, cin, cout = plan[li]
            below = plan[li - 1] if li > 0 else None
            if kind == "conv" and pooled_from == name:
                # g is the gradient wrt the pool output: the pool adjoint and this conv's ReLU mask are folded into
                # the data-gradient's input transform
                pooled_from = None
                p = self.params[name]
                bname, bkind = below[0], below[1]
                assert bkind == "conv"
                g = ops.conv3x3_dgrad_pool(g, acts[name], p["dgrad"], cin, x_in=acts[bname],
                                           addend=style_grads.get(bname), relu_bits=getattr(acts, "relu_bits", {}).get(name),
                                           hw=getattr(acts, "hw", {}).get(name),
                                           addend_unmasked=bname in unmasked and bname in style_grads)
                continue
            if kind == "conv":
                p = self.params[name]
                if below is None:
                    return ops.conv3x3_dgrad(g, p["dgrad"], cin)
                bname, bkind = below[0], below[1]
                if bkind == "conv":
                    # below is a post-ReLU conv output: fold its ReLU mask and its style gradient
                    g = ops.conv3x3_dgrad(g, p["dgrad"], cin, x_in=acts[bname], addend=style_grads.get(bname),
                                          relu_bits=getattr(acts, "relu_bits", {}).get(name),
                                          addend_unmasked=bname in unmasked and bname in style_grads)
                else:
                    g = ops.conv3x3_dgrad(g, p["dgrad"], cin)   # gradient wrt the pooled tensor
            else:
                bname = below[0]                                 # the conv feeding this pool
                xb = acts[bname]                                 # None: a pooled layer whose output was not kept
                hw = (xb.shape[1], xb.shape[2]) if xb is not None else getattr(acts, "hw", {})[bname]
                b2 = plan[li - 2] if li > 1 else None
                if (style_grads.get(bname) is None and b2 is not None and b2[1] == "conv" and below[2] % 64 == 0
                        and hw[0] >= 2 and hw[1] >= 2):
                    pooled_from = bname                          # keep g pooled; fused into the conv below
                else:
                    assert xb is not None, "the output of %s is needed here: list it in ``keep``" % bname
                    g = ops.avgpool2_bwd(g, xb.shape, x=xb, addend=style_grads.get(bname))
        raise AssertionError("unreachable")


def load_vgg(model_path, device, seed=123, synthetic=None):
    """Counterpart of vgg.load_vgg (vgg.py:110-120).  ``model_path`` '.../vgg_19.ckpt' -> the weights converted
    offline to '.../vgg_19.npz' (TensorFlow checkpoints cannot be read here).  Without that file the loader RAISES --
    a stylisation against random filters looks plausible and means nothing -- unless seeded synthetic He-normal
    weights are asked for explicitly: ``synthetic=True`` (config.synthetic_weights) or NFS_SYNTHETIC_VGG=1 (what the
    tests and the benchmark use: no checkpoint exists offline).  ``net.source`` names what was loaded."""
    name = os.path.basename(model_path).split(".")[0]
    blocks = VGG16_BLOCKS if "16" in name else VGG19_BLOCKS
    scope = "vgg_16" if "16" in name else "vgg_19"
    npz = os.path.splitext(model_path)[0] + ".npz"
    if synthetic is None:
        synthetic = os.environ.get("NFS_SYNTHETIC_VGG", "0") == "1"
    if os.path.exists(npz):
        w = load_npz_weights(npz, blocks, scope)
        src = npz
    elif synthetic:
        w = synthetic_weights(seed, blocks)
        src = "synthetic(seed=%d)" % seed
    else:
        raise FileNotFoundError(
            "%s not found%s.  Convert the slim checkpoint to an .npz with keys 'conv1_1/weights' [3,3,Cin,Cout], "
            "'conv1_1/biases', ... (INTEGRATION.md), or opt in to seeded SYNTHETIC weights with config.synthetic_weights"
            "=True / NFS_SYNTHETIC_VGG=1 (results are then not a stylisation by VGG-19)"
            % (npz, " (a TensorFlow checkpoint %s is present but cannot be read without TensorFlow)" % model_path
               if os.path.exists(model_path) or os.path.exists(model_path + ".index") else ""))
    net = VGG(w, device, blocks)
    net.source = src
    print("loss network: %s weights from %s" % (scope, src), file=sys.stderr)
    return net
