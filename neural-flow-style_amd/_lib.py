"""ctypes loader for libnfs_hip.so (the C ABI declared in include/nfs_hip.h).

The HIP library is the product path: there is NO CPU fallback.  If the shared
object is missing, importing an op raises immediately (``NfsLibraryError``).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# NFS_LIB_PATH: load another build of the same library (measurement builds made with -DNFS_ABLATE; never the product)
LIB_PATH = os.environ.get("NFS_LIB_PATH") or os.path.join(_HERE, "libnfs_hip.so")
CSRC_DIR = os.path.join(_HERE, "csrc")


class NfsLibraryError(RuntimeError):
    pass


NFS_EINVAL, NFS_ELAUNCH = -1, -2            # include/nfs_hip.h


class NfsError(RuntimeError):
    """an entry point returned non-zero: ``code`` is its NFS_E* value, the message carries nfs_last_error()"""

    def __init__(self, name, code, msg):
        RuntimeError.__init__(self, "%s failed (%d): %s" % (name, code, msg))
        self.name, self.code = name, int(code)


class SplatCfg(C.Structure):
    _fields_ = [("nd", C.c_int), ("res", C.c_int * 3), ("domain", C.c_float * 3),
                ("radius", C.c_float), ("support", C.c_float), ("rest_density", C.c_float),
                ("nsize", C.c_int), ("clip", C.c_int), ("mode", C.c_int)]


class GramLayer(C.Structure):
    """nfs_gram_layer_t (include/nfs_hip.h): one style layer of the grouped Gram / style-loss entry points"""
    _fields_ = [("F", C.c_void_p), ("Gs", C.c_void_p), ("G", C.c_void_p), ("Dmat", C.c_void_p), ("dF", C.c_void_p),
                ("B", C.c_int), ("Bs", C.c_int), ("HW", C.c_int), ("C", C.c_int),
                ("scale", C.c_float), ("weight", C.c_float), ("relu_mask", C.c_int)]


class ConvDesc(C.Structure):
    """nfs_conv2d_desc_t (include/nfs_hip.h): one convolution of a grouped launch"""
    _fields_ = [("x", C.c_void_p), ("x_mask", C.c_void_p), ("packed", C.c_void_p), ("bias", C.c_void_p),
                ("y", C.c_void_p), ("y_pre", C.c_void_p),
                ("ldx", C.c_int), ("ldm", C.c_int), ("ldy", C.c_int), ("ldp", C.c_int),
                ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int), ("kh", C.c_int), ("kw", C.c_int),
                ("relu", C.c_int), ("accumulate", C.c_int), ("sum_with_prev", C.c_int)]


_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_int64

# name -> argtypes (all return int unless listed in _RESTYPE)
SIGNATURES = {
    "nfs_version": [],
    "nfs_last_error": [],
    "nfs_device_cus": [],
    "nfs_gemm_timer": [_I],
    "nfs_gemm_mode": [_I],
    "nfs_gemm_timer_read": [_P, _P, _P],
    "nfs_gemm_timer_read_kind": [_I, _P, _P, _P, _P],
    "nfs_warp3d_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "nfs_warp3d_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "nfs_rotate_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "nfs_rotate_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P],
    "nfs_advect_fwd": [_P, _P, _P, _I, _I, _I, _I, _P],
    "nfs_advect_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "nfs_advect_bwd_adam": [_P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _F, _P],
    "nfs_advect_fwd_slab": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "nfs_advect_bwd_adam_slab": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _F, _F, _P],
    "nfs_advect_bwd_adam_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _F, _P],
    "nfs_advect_bwd_adam_fwd_slab": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _F, _F, _P],
    "nfs_warp2d_fwd": [_P, _P, _P, _I, _I, _I, _I, _P],
    "nfs_warp2d_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "nfs_advect2d_fwd": [_P, _P, _P, _I, _I, _I, _P],
    "nfs_advect2d_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "nfs_advect_maccormack": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "nfs_curl_fwd": [_P, _P, _I, _I, _I, _I, _P],
    "nfs_curl_bwd": [_P, _P, _I, _I, _I, _I, _P],
    "nfs_lap_down": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "nfs_lap_up": [_P, _P, _F, _P, _P, _I, _I, _I, _I, _I, _P],
    "nfs_lap_up_rms_parts": [_I, _I, _I, _I, _I],
    "nfs_lap_up_rms": [_P, _P, _F, _P, _P, _I, _L, _F, _P, _P, _I, _I, _I, _I, _I, _P],
    "nfs_normalize_mean": [_P, _P, _L, _I, _F, _P, _I, _P],
    "nfs_transport_step": [_P, _P, _F, _F, _P, _F, _P, _I, _I, _I, _I, _P],
    "nfs_smooth3d_relu_fwd": [_P, _P, _I, _I, _I, _F, _P],
    "nfs_smooth3d_relu_bwd": [_P, _P, _P, _I, _I, _I, _F, _P],
    "nfs_render_fwd": [_P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "nfs_render_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P, _P],
    "nfs_rotate_render_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "nfs_render_coef_layout": [_I, _I, _I, _I, _P, _P],
    "nfs_rotate_render_fwd_coef": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "nfs_render_ray_coef": [_P, _P, _P, _P, _I, _I, _I, _F, _P],
    "nfs_render_ray_coef_bounds": [_I, _I, _I],
    "nfs_rotate_bwd_coef": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P],
    "nfs_live_mask_words": [_I, _I, _I],
    "nfs_advect_fwd_live": [_P, _P, _P, _P, _I, _I, _I, _P],
    "nfs_advect_bwd_adam_fwd_live": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _F, _P],
    "nfs_advect_bwd_adam_fwd_live_ever": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _F, _P],
    "nfs_rotate_bwd_coef_live": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P, _I, _P, _P],
    "nfs_rotate_live_workspace_ints": [_I, _I, _I],
    "nfs_rotate_render_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "nfs_maxnorm_fwd": [_P, _P, _P, _I, _I, _P],
    "nfs_maxnorm_bwd": [_P, _P, _P, _P, _I, _I, _P, _P],
    "nfs_maxnorm_input_fwd": [_P, _P, _P, _I, _I, _P],
    "nfs_maxnorm_input_bwd": [_P, _P, _P, _P, _I, _I, _P, _P],
    "nfs_loss_net_input_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "nfs_loss_net_input_bwd": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "nfs_conv3x3_packed_floats": [_I, _I, _I],
    "nfs_conv3x3_pack": [_P, _P, _I, _I, _I, _P],
    "nfs_conv3x3_workspace_floats": [_I, _I, _I, _I, _I],
    "nfs_conv3x3_relu_bits_words": [_I, _I, _I, _I, _I, _I],
    "nfs_conv3x3_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _L, _P, _P],
    "nfs_conv3x3_dgrad": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _L, _P, _I, _P],
    "nfs_conv3x3_fwd_pool": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _L, _P, _P],
    "nfs_conv3x3_dgrad_pool": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _L, _P, _I, _P],
    "nfs_avgpool2_fwd": [_P, _P, _I, _I, _I, _I, _P],
    "nfs_avgpool2_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "nfs_gram_workspace_floats": [_I, _I, _I],
    "nfs_gram_fwd": [_P, _P, _I, _I, _I, _P, _F, _P, _L, _P],
    "nfs_style_loss_fwd": [_P, _P, _P, _P, _I, _I, _I, _F, _P],
    "nfs_content_loss": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _P],
    "nfs_content_loss_signed": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _P],
    "nfs_gram_bwd": [_P, _P, _P, _I, _I, _I, _P, _F, _I, _P],
    "nfs_hist_loss": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "nfs_conv3x3_executed_flops": [_I, _I, _I, _I, _I, _I],
    "nfs_conv2d_packed_floats": [_I, _I, _I, _I, _I],
    "nfs_conv2d_pack": [_P, _P, _I, _I, _I, _I, _I, _P],
    "nfs_conv2d_workspace_floats": [_I, _I, _I, _I, _I, _I, _I, _I],
    "nfs_conv2d_fwd": [_P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _L, _P],
    "nfs_conv2d_group_workspace_floats": [_P, _I, _I],
    "nfs_conv2d_group": [_P, _I, _I, _P, _L, _P],
    "nfs_conv2d_dgrad_small": [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "nfs_maxpool3_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "nfs_maxpool3_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P],
    "nfs_lrn_fwd": [_P, _P, _P, _L, _I, _I, _I, _F, _F, _F, _P],
    "nfs_lrn_bwd": [_P, _P, _P, _P, _P, _L, _I, _I, _I, _F, _F, _I, _P],
    "nfs_avgpool_valid_fwd": [_P, _P, _I, _I, _I, _I, _I, _P],
    "nfs_avgpool_valid_bwd": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "nfs_relu_mask_add": [_P, _I, _P, _I, _P, _I, _P, _I, _L, _I, _P],
    "nfs_gram_style_group_workspace_floats": [_P, _I],
    "nfs_gram_style_group_parts": [_P, _I],
    "nfs_gram_style_group_fwd": [_P, _I, _P, _P, _L, _P],
    "nfs_gram_group_bwd": [_P, _I, _P],
    "nfs_hist_loss_masked": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "nfs_hist_loss_wide_workspace_floats": [_I, _I, _I, _I],
    "nfs_hist_loss_wide": [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _F, _I, _P],
    "nfs_resize_bicubic_tf1": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "nfs_style_mask_apply": [_P, _P, _P, _P, _I, _I, _I, _P],
    "nfs_style_mask_bwd": [_P, _P, _P, _P, _I, _I, _I, _P],
    "nfs_tv_loss": [_P, _P, _P, _I, _I, _I, _I, _F, _P],
    "nfs_p2g_fwd": [_P, _P, _P, _P, _P, _I, _I, C.POINTER(SplatCfg), _P],
    "nfs_p2g_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, C.POINTER(SplatCfg), _P],
    "nfs_p2g_wavg_finish": [_P, _P, _P, _L, _I, _F, _P],
    "nfs_p2g_wavg_finish_bwd": [_P, _P, _P, _P, _P, _L, _I, _F, _P],
    "nfs_p2g_wavg_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _F, C.POINTER(SplatCfg), _P],
    "nfs_g2p_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _L, _I, _P],
    "nfs_adam_tf_step": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _P],
    "nfs_fill": [_P, _F, _L, _P],
    "nfs_axpy": [_P, _P, _F, _L, _P],
    "nfs_slab_pack": [_P, _P, _I, _L, _I, _I, _P],
    "nfs_colour_clamp_gather": [_P, _P, _P, _L, _I, _P],
    "nfs_clamp01_bwd": [_P, _P, _P, _L, _P],
    "nfs_colour_clamp_scatter_bwd": [_P, _P, _P, _P, _L, _I, _P],
    "nfs_iterate_update": [_P, _P, _L, _P],
}
_RESTYPE = {"nfs_last_error": C.c_char_p, "nfs_conv3x3_packed_floats": C.c_int64,
            "nfs_conv3x3_workspace_floats": C.c_int64, "nfs_gram_workspace_floats": C.c_int64,
            "nfs_conv3x3_relu_bits_words": C.c_int64, "nfs_gram_style_group_workspace_floats": C.c_int64, "nfs_hist_loss_wide_workspace_floats": C.c_int64,
            "nfs_conv3x3_executed_flops": C.c_double, "nfs_conv2d_packed_floats": C.c_int64,
            "nfs_conv2d_workspace_floats": C.c_int64, "nfs_conv2d_group_workspace_floats": C.c_int64}

_lib = None
ABI_VERSION = 151          # nfs_version() this table was written against (include/nfs_hip.h)


def build(verbose=False):
    """Compile csrc/*.hip for gfx950 into libnfs_hip.so (in-tree)."""
    r = subprocess.run(["make", "-C", CSRC_DIR, "-j", str(os.cpu_count() or 4)],
                       capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode != 0:
        raise NfsLibraryError("building libnfs_hip.so failed")
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NfsLibraryError(
                "libnfs_hip.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                "-- there is no CPU fallback for the product path" % LIB_PATH)
        # PyTorch-ROCm ships its own HIP runtime; it must be the one already resident when libnfs_hip.so resolves
        # libamdhip64 -- loading this library first leaves the process with two runtimes and every later launch
        # fails with "no ROCm-capable device is detected" (seen when build() and smoke() shared a process)
        import torch  # noqa: F401
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise NfsLibraryError("cannot load %s: %s" % (LIB_PATH, e))
        for name, argt in SIGNATURES.items():
            try:
                f = getattr(L, name)
            except AttributeError:
                raise NfsLibraryError("libnfs_hip.so does not export %s (stale build?)" % name)
            f.argtypes = argt
            f.restype = _RESTYPE.get(name, C.c_int)
        # the binding and the library move together (entry points, packed-filter sizes, the default GEMM arithmetic): a
        # stale .so with every symbol of an older minor present must not be driven by this table
        if L.nfs_version() < ABI_VERSION:
            raise NfsLibraryError("libnfs_hip.so reports ABI %d, this binding needs >= %d: rebuild (make -C %s)"
                                  % (L.nfs_version(), ABI_VERSION, CSRC_DIR))
        _lib = L
    return _lib


# Optional live profiling (used by bench.py): when PROFILE is a dict, every call is bracketed by
# two events recorded on the SAME stream the kernel is enqueued on; entries are
# name -> [(start_event, end_event, args)].  Costs two event records per call, so it is off by
# default and the headline timing in bench.py is taken with it off.
PROFILE = None


def call(name, *args):
    """Call an int-returning entry point; raise with nfs_last_error() on failure."""
    L = lib()
    if PROFILE is not None:
        import torch
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(L, name)(*args)
        e1.record()
        PROFILE.setdefault(name, []).append((e0, e1, args))
    else:
        rc = getattr(L, name)(*args)
    if rc != 0:
        raise NfsError(name, rc, L.nfs_last_error().decode())
    return rc
