"""Dead-region skipping for the velocity variable (nfs_advect_*_live + nfs_rotate_bwd_coef_live).

The adjoint of advect (transform.py:557-569) w.r.t. the velocity is g(x) * grad d0(x - v): an exact zero wherever the
eight corners of the back-traced stencil are equal, whatever g(x) is.  The forward advect writes the mask of the voxels
where they differ; the rotate adjoint leaves dL/d d_s unsummed where only masked-out voxels would read it.  What must
hold: (1) the mask is sound (no voxel with a non-zero velocity gradient is masked out) and tight, (2) the rotate adjoint
with a mask is bit-identical to the one without on every voxel within the stencil reach of a live voxel, (3) whole
optimisation steps -- variable and Adam moments -- are BIT-identical with skipping on and off, on a smoke-like density
(tiles skipped) and on a dense one (nothing skipped)."""
import numpy as np
import pytest
import torch

from tests.synth import blob_density, style_image, uniform_views

pytestmark = pytest.mark.gpu


def _bits(mask_f32, n):
    """live mask buffer -> bool [n]"""
    words = mask_f32.view(torch.int64).cpu().numpy().view(np.uint64)
    b = np.unpackbits(words.view(np.uint8), bitorder="little")
    return b[:n].astype(bool)


def _dilate(m, r):
    """box dilation of a bool [D,H,W] array by r cells"""
    if r == 0:
        return m
    t = torch.tensor(m[None, None].astype(np.float32))
    return (torch.nn.functional.max_pool3d(t, 2 * r + 1, 1, r)[0, 0] > 0).numpy()


@pytest.mark.parametrize("shape", [(24, 20, 28), (17, 12, 36)])
def test_live_mask_is_sound_and_tight(shape):
    import neural_flow_style_amd.ops as ops
    D, H, W = shape
    rng = np.random.RandomState(5)
    # a density with empty space, a plateau (clipped at 1) and smooth flanks
    zz, yy, xx = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij")
    r2 = ((zz - D / 2) / (D / 3.0)) ** 2 + ((yy - H / 2) / (H / 3.0)) ** 2 + ((xx - W / 2) / (W / 3.0)) ** 2
    d = np.clip(2.5 * np.exp(-3.0 * r2), 0, 1).astype(np.float32)
    d[d < 0.05] = 0.0
    vel = (rng.randn(D, H, W, 3) * 1.5 / (max(shape) - 1)).astype(np.float32)
    dg = torch.tensor(d).cuda().unsqueeze(-1)
    vg = torch.tensor(vel).cuda()
    live = ops.live_mask(D, H, W, dg)
    out_live = ops.advect_fwd(dg, vg, live=live)
    out_ref = ops.advect_fwd(dg, vg)
    assert torch.equal(out_live, out_ref)                      # the sample itself does not change
    m = _bits(live, D * H * W).reshape(D, H, W)
    g = torch.tensor(rng.randn(D, H, W, 1).astype(np.float32)).cuda()
    _, g_vel = ops.advect_bwd(dg, vg, g, need_d=False, need_vel=True)
    nz = (g_vel != 0).any(-1).cpu().numpy()
    assert not (nz & ~m).any(), "a voxel with a non-zero velocity gradient is masked out"
    assert 0.05 < m.mean() < 0.9                               # empty space and the plateau are dead, the flanks live
    # tight: next to none of the live voxels has an all-zero gradient for a random g (clamped-out voxels aside)
    assert (m & ~nz).sum() <= 0.02 * m.sum() + 8


@pytest.mark.parametrize("dilate", [0, 1])
@pytest.mark.parametrize("shape,V", [((30, 30, 70), 3), ((16, 33, 20), 2)])
def test_rotate_adjoint_with_mask_is_bit_identical_where_it_is_read(shape, V, dilate):
    import neural_flow_style_amd.ops as ops
    import neural_flow_style_amd.transform as T
    D, H, W = shape
    rng = np.random.RandomState(11)
    d = torch.tensor(rng.rand(D, H, W).astype(np.float32)).cuda()
    rot = T.rot_to_device(uniform_views(V), "cuda")
    u_rot = torch.empty((V, D, H, W), dtype=torch.float32, device="cuda")
    img, rs, _, seg = ops.rotate_render_fwd_coef(d, rot, 0.05, u_rot=u_rot)
    g_img = torch.tensor(rng.randn(V, H, W).astype(np.float32)).cuda()
    ab, bounds = ops.render_ray_coef(g_img, seg, 0.05)
    full = ops.rotate_bwd_coef(u_rot, ab, rot, bounds)
    # live voxels: a blob off-centre (so that whole tiles are dead and others are cut by its box) + one lone voxel
    zz, yy, xx = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij")
    m = ((zz - 0.3 * D) ** 2 + (yy - 0.6 * H) ** 2 + (xx - 0.25 * W) ** 2) < (0.2 * min(shape)) ** 2
    m[D - 1, 0, W - 1] = True
    words = np.zeros(int(ops._lib.lib().nfs_live_mask_words(D, H, W)) * 8, np.uint8)
    packed = np.packbits(m.reshape(-1), bitorder="little")
    words[:packed.size] = packed
    live = torch.tensor(words.view(np.float32)).cuda()
    for overwrite in (True, False):
        if overwrite:
            got = ops.rotate_bwd_coef(u_rot, ab, rot, bounds, live=live, dilate=dilate)
            base = full
        else:
            acc0 = torch.tensor(rng.randn(D, H, W).astype(np.float32)).cuda()
            got = ops.rotate_bwd_coef(u_rot, ab, rot, bounds, g_d_acc=acc0.clone(), overwrite=False, live=live,
                                      dilate=dilate)
            base = ops.rotate_bwd_coef(u_rot, ab, rot, bounds, g_d_acc=acc0.clone(), overwrite=False)
        need = torch.tensor(_dilate(m, dilate)).cuda()
        assert torch.equal(got[need], base[need])
        assert torch.isfinite(got).all()
        if overwrite:
            # far from every live voxel nothing was summed
            far = ~torch.tensor(_dilate(m, dilate + 40)).cuda()
            assert (got[far] == 0).all()
    # an empty mask: every tile returns at once
    z = ops.rotate_bwd_coef(u_rot, ab, rot, bounds, live=torch.zeros_like(live), dilate=dilate)
    assert (z == 0).all()
    # a full mask: the same sums everywhere
    ones = torch.tensor(np.full(words.size, 0xFF, np.uint8).view(np.float32)).cuda()
    assert torch.equal(ops.rotate_bwd_coef(u_rot, ab, rot, bounds, live=ones, dilate=dilate), full)


def _stylizer(G, V, density, skip, graph, seed=3):
    import neural_flow_style_amd.engine as eng
    import neural_flow_style_amd.transform as T
    import neural_flow_style_amd.vgg as vgg
    rng = np.random.RandomState(seed)
    d0 = blob_density(G, rng)
    if density == "dense":
        d0 = (0.2 + 0.6 * rng.rand(G, G, G)).astype(np.float32)
    vel0 = (rng.randn(G, G, G, 3) * 0.3 / (G - 1)).astype(np.float32)
    layers = ["conv1_1", "conv2_1", "conv3_1"]
    net = vgg.VGG(vgg.synthetic_weights(123, upto="conv3_1"), "cuda")
    loss = eng.RenderStyleLoss(net, layers, [1.0] * 3, 1.0, transmit=0.05)
    loss.set_style_image(style_image(G, G, rng))
    gs = eng.GridStylizer(loss, torch.tensor(d0).cuda(), k=3, target="v", lr=2e-3, graph=graph)
    gs.dead_skip = skip
    gs.var.copy_(torch.tensor(vel0))
    return gs, T.rot_to_device(uniform_views(V), "cuda")


@pytest.mark.parametrize("density", ["smoke", "dense"])
@pytest.mark.parametrize("graph", [False, True])
def test_steps_are_bit_identical_with_skipping_on_and_off(density, graph):
    G, V, K = 44, 2, 4
    res = {}
    for skip in (False, True):
        gs, rot = _stylizer(G, V, density, skip, graph)
        losses = [float(gs.step(rot)) for _ in range(K)]
        res[skip] = (gs.var.clone(), gs.adam.m.clone(), gs.adam.v.clone(), losses, gs)
    a, b = res[False], res[True]
    assert a[3] == b[3]                                        # the losses (computed before the adjoint) agree exactly
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)
    gs_on = b[4]
    assert gs_on._live_valid() and gs_on._live_kw(), "the run with skipping on never had a mask"
    frac = _bits(gs_on._live_buf, G ** 3).mean()
    if density == "dense":
        assert frac > 0.99                                     # nothing to skip: the control case
        # ... and then dL/d d_s itself is the same everywhere
        assert torch.equal(a[4].g_ds, gs_on.g_ds)
    else:
        assert frac < 0.7
        assert not torch.equal(a[4].g_ds, gs_on.g_ds)          # tiles were skipped (else this test shows nothing)


def test_gradient_and_field_gradient_respect_the_contract():
    """gradient() may skip (it hands back the VARIABLE's gradient); field_gradient() called by itself returns the full
    dL/d d_s, mask or not; the density variable never skips"""
    G, V = 40, 2
    gs_off, rot = _stylizer(G, V, "smoke", False, False)
    gs_on, _ = _stylizer(G, V, "smoke", True, False)
    _, g_off = gs_off.gradient(rot)
    _, g_on = gs_on.gradient(rot)
    assert torch.equal(g_off, g_on)
    _, gds_on = gs_on.field_gradient(rot)
    _, gds_off = gs_off.field_gradient(rot)
    assert torch.equal(gds_on, gds_off)
    import neural_flow_style_amd.engine as eng
    gd = eng.GridStylizer(gs_on.loss, gs_on.d0, k=3, target="d")
    assert gd._live_target() is None and not gd._live_kw()


def test_mask_follows_the_variable():
    """a velocity set by hand, a re-bound frame: the mask is rebuilt with the forward advect, never reused stale"""
    G, V = 40, 2
    gs, rot = _stylizer(G, V, "smoke", True, False)
    gs.step(rot)
    assert gs._live_valid()
    gs.var.mul_(0.5)                                           # in-place torch op: version counter moves
    assert not gs._live_valid()
    ref, _ = _stylizer(G, V, "smoke", False, False)
    ref.step(rot)
    ref.var.mul_(0.5)
    gs.step(rot); ref.step(rot)
    assert torch.equal(gs.var, ref.var)
    rng = np.random.RandomState(9)
    d1 = torch.tensor(blob_density(G, rng)).cuda()
    gs.bind(d1); ref.bind(d1)
    gs.step(rot); ref.step(rot)
    assert torch.equal(gs.var, ref.var)


def test_more_views_than_one_launch_takes():
    """33 views = two launches of the adjoint (32 + 1) over ONE set of boxes: the first writes (overwrite), the second
    accumulates inside the boxes only; needed voxels bit-equal to the unmasked pair of launches"""
    import neural_flow_style_amd.ops as ops
    import neural_flow_style_amd.transform as T
    D, H, W, V = 16, 18, 40, 33
    rng = np.random.RandomState(17)
    d = torch.tensor(rng.rand(D, H, W).astype(np.float32)).cuda()
    rot = T.rot_to_device(uniform_views(V), "cuda")
    u_rot = torch.empty((V, D, H, W), dtype=torch.float32, device="cuda")
    _, _, _, seg = ops.rotate_render_fwd_coef(d, rot, 0.05, u_rot=u_rot)
    ab, bounds = ops.render_ray_coef(torch.tensor(rng.randn(V, H, W).astype(np.float32)).cuda(), seg, 0.05)
    full = ops.rotate_bwd_coef(u_rot, ab, rot, bounds)
    m = np.zeros((D, H, W), bool)
    m[3:9, 10:15, 5:12] = True
    words = np.zeros(int(ops._lib.lib().nfs_live_mask_words(D, H, W)) * 8, np.uint8)
    packed = np.packbits(m.reshape(-1), bitorder="little")
    words[:packed.size] = packed
    live = torch.tensor(words.view(np.float32)).cuda()
    got = ops.rotate_bwd_coef(u_rot, ab, rot, bounds, live=live, dilate=1)
    need = torch.tensor(_dilate(m, 1)).cuda()
    assert torch.equal(got[need], full[need]) and torch.isfinite(got).all()


def test_view_groups_on_side_streams_and_no_smoothing():
    """view groups whose chains run on separate streams use one box workspace per stream (the adjoints run concurrently);
    k = 0 (no smoothing between the rotate and the advect adjoint) uses a reach of 0: both bit-equal to skipping off"""
    import neural_flow_style_amd.engine as eng
    for groups, k in ((2, 3.0), (1, 0.0)):
        res = {}
        for skip in (False, True):
            gs, rot = _stylizer(44, 4, "smoke", skip, False)
            gs.k = k
            gs.loss.view_groups = groups
            gs.loss.vgg_streams = groups
            for _ in range(3):
                gs.step(rot)
            res[skip] = (gs.var.clone(), gs.adam.m.clone(), gs.adam.v.clone(), bool(gs._live_kw()),
                         gs._live_kw().get("dilate"))
        assert res[True][3] and res[True][4] == (1 if k > 0 else 0)
        for x, y in zip(res[True][:3], res[False][:3]):
            assert torch.equal(x, y), (groups, k)


def test_never_live_waves_are_left_out_of_the_adam_kernel_only_while_the_moments_are_its_own():
    """the fused advect-adjoint + ApplyAdam kernel skips the 256-voxel waves none of whose voxels has ever been live
    (TFAdamState.ever_mask).  That rests on m = v = +0 there, so the mask is dropped as soon as anything else writes the
    moments -- the plain Adam step, a copy into m -- and the steps stay bit-identical to skipping off throughout"""
    G, V = 44, 2
    gs, rot = _stylizer(G, V, "smoke", True, False)
    ref, _ = _stylizer(G, V, "smoke", False, False)
    for _ in range(2):
        gs.step(rot); ref.step(rot)
    ev = gs.adam.ever_mask()
    assert ev is not None and ref.adam.ever_mask() is None
    ever = _bits(ev, G ** 3)
    live = _bits(gs._live_buf, G ** 3)
    assert 0.05 < ever.mean() < 0.8                            # (the current mask is OR-ed in by the next step's kernel)
    del live
    # outside the mask the moments are exactly zero
    m_nz = (gs.adam.m != 0).any(-1).reshape(-1).cpu().numpy() | (gs.adam.v != 0).any(-1).reshape(-1).cpu().numpy()
    assert not (m_nz & ~ever).any()
    # a foreign write to the moments (here: a no-op in value, but the version counter moves): the mask is dropped
    gs.adam.m.mul_(1.0); ref.adam.m.mul_(1.0)
    assert gs.adam.ever_mask() is None
    for _ in range(2):
        gs.step(rot); ref.step(rot)
    assert gs.adam.ever_mask() is None                         # ... and does not come back for this state
    assert torch.equal(gs.var, ref.var) and torch.equal(gs.adam.m, ref.adam.m) and torch.equal(gs.adam.v, ref.adam.v)
    # a fresh state on the same stylizer starts a fresh mask
    import neural_flow_style_amd.engine as eng
    gs.adam = eng.TFAdamState(); ref.adam = eng.TFAdamState()
    for _ in range(2):
        gs.step(rot); ref.step(rot)
    assert gs.adam.ever_mask() is not None
    assert torch.equal(gs.var, ref.var) and torch.equal(gs.adam.m, ref.adam.m) and torch.equal(gs.adam.v, ref.adam.v)
