"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol the
header declares, and the product path refuses to run without the HIP extension / a GPU."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ensure_built():
    from neural_flow_style_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib


def test_library_exports_every_declared_symbol():
    _lib = _ensure_built()
    hdr = open(os.path.join(ROOT, "include", "nfs_hip.h")).read()
    declared = set(re.findall(r"\b(nfs_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.nfs_version() >= 100


def test_argument_errors_do_not_need_a_gpu():
    _lib = _ensure_built()
    with pytest.raises(RuntimeError, match="null pointer"):
        _lib.call("nfs_render_fwd", None, None, None, 1, 1, 1, 1, 0.1, 0, None)
    assert _lib.lib().nfs_conv3x3_packed_floats(3, 64, 0) == 9 * 3 * 64              # conv1_1: direct only
    # + Winograd F(4x4,3x3) filters (36 floats per (ci, co))
    # + the same filters in MFMA fragment order, for the 32x32x2 and the 16x16x4 instruction (36 + 36)
    # + the 16x16x4 pack as three bf16 limb planes for the split-limb GEMM (6 bytes per value: 54)
    # + where both channel counts are >= 128: the F(5x5,3x3) filters, their fragment order (49 + 49) and its limb planes (73.5)
    assert _lib.lib().nfs_conv3x3_packed_floats(256, 512, 0) == (9 + 36 + 36 + 36 + 54 + 98) * 256 * 512 + 147 * 256 * 512 // 2
    # narrow layers (64 / 128 channels both sides): + the filters in the fragment order of the single-kernel path
    assert _lib.lib().nfs_conv3x3_packed_floats(64, 128, 0) == (9 + 36 + 36 + 36 + 54 + 36) * 64 * 128


def test_no_cpu_fallback():
    import neural_flow_style_amd.ops as ops
    with pytest.raises(ValueError):
        ops.render_fwd(torch.zeros(1, 2, 2, 2), 0.1)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "neural-flow-style_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), os.path.join(dirpath, f)


def test_view_sampling_shapes_and_determinism():
    import neural_flow_style_amd.transform as T
    mats, views = T.rot_mat(-5, 5, 5, -10, 10, 10, sample_type="uniform")
    assert len(mats) == 9 and views[0] == {"phi": -5.0, "theta": -10.0}
    # R = Ry(theta) @ Rz(phi), orthonormal
    for m in mats:
        np.testing.assert_allclose(m @ m.T, np.eye(3), atol=1e-12)
    np.testing.assert_allclose(mats[4], np.eye(3), atol=1e-12)
    a, va = T.rot_mat(-5, 5, 5, -10, 10, 10, "poisson", np.random.RandomState(123), nv=8)
    b, vb = T.rot_mat(-5, 5, 5, -10, 10, 10, "poisson", np.random.RandomState(123), nv=8)
    assert len(a) == 8 and va == vb
    for v in va[:-1]:
        assert -5 <= v["phi"] <= 5 and -10 <= v["theta"] <= 10
    # Poisson-disc property: pairwise distance >= r = max(units)/2
    pts = np.array([[v["theta"], v["phi"]] for v in T.rot_mat_poisson(-5, 5, 5, -10, 10, 10, np.random.RandomState(1))])
    dist = np.linalg.norm(pts[:, None] - pts[None], axis=-1) + np.eye(len(pts)) * 1e9
    assert dist.min() >= 5.0 - 1e-9


def test_config_surface_matches_the_reference_defaults():
    """every flag of the reference's config.get_config() (fixture generated from the reference itself by
    tests/golden/make_config_fixture.py) exists with the same default; the build adds only documented extras"""
    import json
    import os
    from neural_flow_style_amd.config import get_config
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_defaults.json")))
    mine = vars(get_config([])[0])
    assert len(ref) >= 60
    for k, v in ref.items():
        assert k in mine, k
        got = mine[k]
        assert (list(got) if isinstance(got, (list, tuple)) else got) == v, (k, got, v)
    assert set(mine) - set(ref) == {"views_mode", "grid_variable", "synthetic_weights", "transport_recursive", "ray_mode"}


def test_product_library_has_no_wrong_result_ablation_switches():
    """the timing-only ablation branches (NFS_GEMM_DBG / NFS_CONV_DBG: skip loads, stores or MFMAs, wrong results by
    construction) exist only in ``make ABLATE=1`` builds: the product library must not even contain the variable names"""
    import os
    from neural_flow_style_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"NFS_GEMM_DBG", b"NFS_CONV_DBG"):
        assert name not in blob, name
