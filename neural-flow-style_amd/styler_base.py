"""Host-side mirror of the reference's ``StylerBase`` (styler_base.py:11-346).

Same attribute surface: every ``config`` attribute is copied onto the object
(styler_base.py:14-15); ``load_img(hw)`` loads / tiles / crops the style image
(311-346); the loss terms of ``_loss`` that the BASELINE configurations use (style
152-185, TV 211-213, pressure 228-230) are evaluated by ``engine.RenderStyleLoss`` on
the HIP kernels, and so is the content term (135-150) on a layer of the same VGG network.
The histogram term (187-209, util.histogram_match_tf) runs on ``nfs_hist_loss``.  The Inception-pb network is out
of scope (its weights are not available) and raises.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import engine, ops, util
from . import transform as T
from . import vgg as vggmod
from .config import complete


class StylerBase(object):
    def __init__(self, self_dict):
        complete(self_dict)
        for arg in vars(self_dict):
            setattr(self, arg, getattr(self_dict, arg))
        if not torch.cuda.is_available():
            raise RuntimeError("the stylizer runs on the HIP kernels only (no CPU fallback): no GPU visible")
        gpu = 0
        if "LOCAL_RANK" in os.environ:
            # one rank per GPU; ranks beyond the device count wrap (gloo functional checks on a one-GPU box)
            gpu = int(os.environ["LOCAL_RANK"]) % max(torch.cuda.device_count(), 1)
        elif str(getattr(self, "gpu_id", "0")).lstrip("-").isdigit() and int(self.gpu_id) >= 0:
            gpu = int(self.gpu_id) % max(torch.cuda.device_count(), 1)
        self.device = torch.device("cuda", gpu)
        torch.cuda.set_device(self.device)
        self.model_path = os.path.join(self.data_dir, self.model_dir, self.network)
        if getattr(self, "w_density", 0) and "d" not in getattr(self, "target_field", ""):
            raise NotImplementedError("the density-preservation loss acts on the particle-density variable "
                                      "(target_field 'd'), as in the reference (styler_3p.py:75)")
        synthetic = True if getattr(self, "synthetic_weights", False) else None
        if "vgg" in self.model_path:
            self.net = vggmod.load_vgg(self.model_path, self.device, seed=getattr(self, "seed", 123), synthetic=synthetic)
        elif "inception" in self.model_path:
            # styler_base.py:17-30: the Inception-v1 GraphDef, its first stride set to 1 with ``pool1``
            from . import inception as incmod
            self.net = incmod.load_inception(self.model_path, self.device, seed=getattr(self, "seed", 123),
                                             synthetic=synthetic, pool1=bool(getattr(self, "pool1", False)))
        else:
            raise NotImplementedError("network=%r: 'vgg_19.ckpt' / 'vgg_16.ckpt' or 'tensorflow_inception_graph.pb'"
                                      % self.network)
        if getattr(self, "w_content", 0):
            # _layer(content_layer) is a lookup among the tensors of the chosen network (styler_base.py:91-94): a name
            # of the other network is an error there too
            names = [s_[0] for s_ in self.net.seq if s_[1] == "conv"]
            if self.content_layer not in names:
                raise KeyError("content_layer %r is not a layer of %s (w_content=%g; pass --w_content 0 for pure "
                               "style transfer as run.bat does)" % (self.content_layer, self.network, self.w_content))
        self.content_img = None
        self.style_img = None

    # -- _plugin_to_loss_net / _loss: built once, evaluated by the engine --------------------------
    def _make_loss(self, rotate, render_liquid=None):
        w_layers = list(self.w_style_layer)
        if len(w_layers) == 1 and len(self.style_layer) > 1:
            w_layers = w_layers * len(self.style_layer)
        return engine.RenderStyleLoss(
            self.net, self.style_layer, w_layers, self.w_style, transmit=self.transmit,
            render_liquid=self.render_liquid if render_liquid is None else render_liquid,
            resize_scale=self.resize_scale, rotate=rotate, w_tv=self.w_tv, v_batch=self.v_batch,
            w_content=getattr(self, "w_content", 0), content_layer=getattr(self, "content_layer", None),
            content_channel=getattr(self, "content_channel", 0), w_content_amp=getattr(self, "w_content_amp", 100),
            w_hist=getattr(self, "w_hist", 0), hist_layer=getattr(self, "hist_layer", ()),
            w_hist_layer=getattr(self, "w_hist_layer", ()), ray_mode=getattr(self, "ray_mode", ""))

    # -- _hist_feature (styler_base.py:280-309): the style image at the loss-net input size -------------------------
    def _hist_feature(self, style_target, style_shp=None):
        """RGBA style images are premultiplied by alpha (283-286); the reference's further masking of the fetched
        features by the resized alpha (299-303) multiplies the graph TENSOR, not the fetched array, i.e. has no effect
        on what is returned -- as with _style_feature."""
        return self._style_feature(style_target, style_shp)

    # -- _content_feature (styler_base.py:232-247): the content image at the loss-net input size -------------------
    def _content_feature(self, content_target, content_shp):
        # top_k > 0 keeps the k strongest classes of Inception's softmax2_pre_activation (styler_base.py:240-245);
        # on any other layer the reference asserts
        assert getattr(self, "top_k", 0) <= 0 or "softmax2_pre_activation" in self.content_layer, \
            "top_k > 0 needs content_layer softmax2_pre_activation (pass --top_k 0 with a VGG content layer)"
        content_target = np.array(content_target, np.float32)
        if not np.isclose(self.resize_scale, 1):
            content_shp = [int(s * self.resize_scale) for s in content_shp]
        if tuple(content_target.shape[:2]) != tuple(int(s) for s in content_shp):
            content_target = util.resize(content_target, content_shp, order=3)
        return content_target

    def _content_top_k(self):
        """styler_base.py:240-245: with a content image, ``top_k`` > 0 keeps the k strongest logits of the fetched
        feature (the flag default is 5, and the reference asserts the content layer is the classifier's logits)"""
        return int(getattr(self, "top_k", 0) or 0)

    # -- _style_feature (styler_base.py:249-278) -----------------------------------------------------
    def _style_feature(self, style_target, style_shp=None):
        style_target = np.array(style_target, np.float32)
        if style_target.shape[-1] == 4:           # RGBA: premultiply by alpha (252-255)
            m = style_target[..., -1] / 255
            style_target = style_target[..., :-1] * np.stack([m] * 3, axis=-1)
        if style_shp is not None:
            if not np.isclose(self.resize_scale, 1):
                style_shp = [int(s * self.resize_scale) for s in style_shp]
            if tuple(style_target.shape[:2]) != tuple(int(s) for s in style_shp):
                style_target = util.resize(style_target, style_shp, order=3)
        return style_target

    # -- load_img (styler_base.py:311-346) -----------------------------------------------------------
    def load_img(self, hw=None):
        from PIL import Image
        self.content_img = None
        self.style_img = None
        content_target = getattr(self, "content_target", "")
        if getattr(self, "w_content", 0) > 0 and (isinstance(content_target, np.ndarray) or bool(content_target)):
            img = np.float32(content_target) if isinstance(content_target, np.ndarray) \
                else np.float32(Image.open(content_target))
            if img.ndim == 2:
                img = np.stack([img] * 3, -1)
            if img.shape[-1] == 4:                           # remove the alpha channel (styler_base.py:317-318)
                img = img[..., :-1]
            if hw is not None:
                img = util.crop_ratio(img, hw[1] / hw[0])
            self.content_img = img
        has_style = isinstance(self.style_target, np.ndarray) or bool(self.style_target)
        if self.w_style > 0 and has_style:
            if isinstance(self.style_target, np.ndarray):
                img = np.float32(self.style_target)          # build extension: in-memory style image
            else:
                img = np.float32(Image.open(self.style_target))
            if img.ndim == 2:
                img = np.stack([img] * 3, -1)
            if self.style_tiling > 1:
                img = np.tile(img, (self.style_tiling, self.style_tiling, 1))
            if hw is not None:
                img = util.crop_ratio(img, hw[1] / hw[0])
            self.style_img = img

    # -- _transport (styler_base.py:59-89): move a grid field from frame a to frame b ----------------
    def _transport(self, g, v, a, b, recursive=True):
        """g [D,H,W,C] device tensor, v [F,D,H,W,3] device tensor (advect units)"""
        if a < b:
            steps = [v[i] for i in range(a, b)] if recursive else [v[a] * (b - a)]
        elif a > b:
            steps = [-v[i] for i in reversed(range(b, a))] if recursive else [-v[a - 1] * (a - b)]
        else:
            steps = []
        for u in steps:
            g = ops.advect_fwd(g.contiguous(), u.contiguous())
        return g
