"""The grid operators beside the hot path: 2-D warp / advect on HIP (with the reference's only known-answer vector run
through the kernel), MacCormack advection, stream-function curl -- each against the oracle restatement."""
import os

import numpy as np
import pytest
import torch

from oracle import nfs_oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    a = torch.as_tensor(np.asarray(a)).double(); b = torch.as_tensor(np.asarray(b)).double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_reference_known_answer_vector_through_the_hip_kernels():
    """transform.py:1859-1885: 5x5 arange image, identity warp returns it, the 'zoom-in' warp (affine 0.5 I) returns
    [[6,6.5,..,8],..,[16,..,18]] -- the only machine-checkable expectation in the reference tree -- through
    nfs_warp2d_fwd and, embedded as a [1,5,5] / [5,1,5] / [5,5,1] volume, through nfs_warp3d_fwd"""
    from neural_flow_style_amd import ops
    z = np.load(os.path.join(GOLD, "warp2d_kat.npz"))
    img, zoom = z["img"].astype(np.float32), z["zoom_in"].astype(np.float32)
    g = O.mgrid(5, 5)                                              # [2,5,5]
    im = torch.tensor(img).reshape(1, 5, 5, 1).cuda()
    ident = ops.warp2d_fwd(im, g[None].contiguous().cuda())
    assert np.array_equal(ident.cpu().numpy().reshape(5, 5), img)
    got = ops.warp2d_fwd(im, (0.5 * g)[None].contiguous().cuda()).cpu().numpy().reshape(5, 5)
    np.testing.assert_allclose(got, zoom, atol=1e-6)
    zero = torch.zeros(5, 5)
    for axes, shape in (((None, 0, 1), (1, 5, 5)), ((0, None, 1), (5, 1, 5)), ((0, 1, None), (5, 5, 1))):
        planes = [zero if a is None else 0.5 * g[a] for a in axes]
        coords = torch.stack(planes).reshape(1, 3, *shape).contiguous().cuda()
        got3 = ops.warp3d_fwd(torch.tensor(img).reshape(1, *shape, 1).cuda(), coords).cpu().numpy().reshape(5, 5)
        np.testing.assert_allclose(got3, zoom, atol=1e-6)


@pytest.mark.parametrize("C", [1, 3])
def test_warp2d_and_advect2d_match_the_oracle_with_gradients(C):
    from neural_flow_style_amd import transform as T
    rng = np.random.RandomState(5)
    B, X, Y = 2, 11, 7
    img = rng.randn(B, X, Y, C).astype(np.float32)
    co = rng.uniform(-1.4, 1.4, (B, 2, X, Y)).astype(np.float32)          # incl. out-of-range (border replicate)
    w = rng.randn(B, X, Y, C).astype(np.float32)
    io, cc = torch.tensor(img, requires_grad=True), torch.tensor(co, requires_grad=True)
    oo = O.batch_warp2d(io, cc, [B, X, Y])
    (oo * torch.tensor(w)).sum().backward()
    ih, ch = torch.tensor(img).cuda().requires_grad_(), torch.tensor(co).cuda().requires_grad_()
    oh = T.batch_warp2d(ih, ch)
    (oh * torch.tensor(w).cuda()).sum().backward()
    assert rel(oh.detach().cpu(), oo.detach()) < 1e-6
    assert rel(ih.grad.cpu(), io.grad) < 1e-5 and rel(ch.grad.cpu(), cc.grad) < 1e-5
    # advect 2-D (transform.py:583-588)
    H, W = 12, 9
    d = rng.rand(1, H, W, C).astype(np.float32)
    v = (rng.randn(1, H, W, 2) * 0.3).astype(np.float32)
    w2 = rng.randn(1, H, W, C).astype(np.float32)
    do, vo = torch.tensor(d, requires_grad=True), torch.tensor(v, requires_grad=True)
    ao = O.advect2d(do, vo)
    (ao * torch.tensor(w2)).sum().backward()
    dh, vh = torch.tensor(d).cuda().requires_grad_(), torch.tensor(v).cuda().requires_grad_()
    ah = T.advect(dh, vh, order=1, is_3d=False)
    (ah * torch.tensor(w2).cuda()).sum().backward()
    assert rel(ah.detach().cpu(), ao.detach()) < 1e-6
    assert rel(dh.grad.cpu(), do.grad) < 1e-5 and rel(vh.grad.cpu(), vo.grad) < 1e-5


def test_maccormack_matches_the_oracle_and_is_bounded():
    from neural_flow_style_amd import synthetic as S
    from neural_flow_style_amd import transform as T
    rng = np.random.RandomState(8)
    G = 18
    d = S.blob_density(G, rng)[None, ..., None]
    v = S.curl_velocity(G, rng, max_cells=2.0)[None]
    want = O.advect_maccormack(torch.tensor(d), torch.tensor(v)).numpy()
    got = T.advect(torch.tensor(d).cuda(), torch.tensor(v).cuda(), order=2, is_3d=True).cpu().numpy()
    # the limiter is a comparison: a voxel whose d_adv sits within rounding of an extremum may fall on either side
    diff = np.abs(got - want)
    assert (diff > 1e-5).mean() < 2e-3 and rel(got, want) < 5e-3
    assert got.min() >= d.min() - 1e-6 and got.max() <= d.max() + 1e-6      # no new extrema
    # sharper than first order: transport a blob there and back
    back = T.advect(torch.tensor(got).cuda(), -torch.tensor(v).cuda(), order=2, is_3d=True).cpu().numpy()
    f1 = T.advect(T.advect(torch.tensor(d).cuda(), torch.tensor(v).cuda()), -torch.tensor(v).cuda()).cpu().numpy()
    assert np.abs(back - d).mean() < np.abs(f1 - d).mean()
    # 2-D, 3 channels
    d2 = rng.rand(1, 20, 14, 3).astype(np.float32)
    v2 = (rng.randn(1, 20, 14, 2) * 0.1).astype(np.float32)
    want2 = O.advect_maccormack(torch.tensor(d2), torch.tensor(v2)).numpy()
    got2 = T.advect(torch.tensor(d2).cuda(), torch.tensor(v2).cuda(), order=2, is_3d=False).cpu().numpy()
    assert (np.abs(got2 - want2) > 1e-5).mean() < 5e-3


def test_curl_matches_the_reference_lines_and_its_adjoint():
    from neural_flow_style_amd import transform as T
    rng = np.random.RandomState(9)
    for shape, is_2d in (((2, 9, 7, 1), True), ((1, 6, 8, 5, 3), False), ((2, 2, 2, 2, 3), False)):
        s = rng.randn(*shape).astype(np.float32)
        so = torch.tensor(s, requires_grad=True)
        co = O.curl(so, is_2d=is_2d)
        w = rng.randn(*co.shape).astype(np.float32)
        (co * torch.tensor(w)).sum().backward()
        sh = torch.tensor(s).cuda().requires_grad_()
        ch = T.curl(sh, is_2d=is_2d)
        (ch * torch.tensor(w).cuda()).sum().backward()
        assert ch.shape == co.shape
        assert rel(ch.detach().cpu(), co.detach()) < 1e-6, shape
        assert rel(sh.grad.cpu(), so.grad) < 1e-6, shape
    # a curl field is discretely divergence-free: forward differences commute, so Dx u + Dy v + Dz w == 0 wherever no
    # replicated last slice is involved (u is the component along x = axis W, v along y = axis H, w along z = axis D)
    s = torch.tensor(rng.randn(1, 10, 11, 12, 3).astype(np.float32)).cuda()
    c = T.curl(s, is_2d=False)[0]
    u, v, w = c[..., 0], c[..., 1], c[..., 2]
    div = (u[:-2, :-2, 1:-1] - u[:-2, :-2, :-2]) + (v[:-2, 1:-1, :-2] - v[:-2, :-2, :-2]) + \
          (w[1:-1, :-2, :-2] - w[:-2, :-2, :-2])
    assert float(div.abs().max()) < 1e-5


def test_laplacian_pyramid_3d_kernels_equal_the_gather_forms():
    """the staged 3-D kernels (lap_down from an LDS tile, lap_up by rows -- a block = one coarse cell row of y, all of x, two
    cells of z, the coarse rows it reaches staged in LDS; taken for 1 and 3 channels from 4096 voxels on) against the plain
    gather kernels (any other channel count): the same taps, equal to the rounding of a 125-term float sum (measured:
    6e-8 / 5e-7 absolute on O(1) data) -- odd sizes, ragged tiles, both SAME-padding parities, volumes of one or two
    blocks per axis"""
    import neural_flow_style_amd.ops as ops
    from neural_flow_style_amd import util
    k = torch.as_tensor(util.lap_kernel(True)).cuda().contiguous()
    for shape in ((131, 129, 130), (128, 130, 127), (21, 19, 37), (40, 18, 34), (17, 16, 16), (16, 17, 65)):
        gen = torch.Generator(device="cuda").manual_seed(7)
        x3 = torch.randn(*shape, 3, device="cuda", generator=gen)
        x2 = x3[..., :2].contiguous()                                   # two channels: the gather kernels
        lo3, lo2 = ops.lap_down(x3, k), ops.lap_down(x2, k)
        assert float((lo3[..., :2] - lo2).abs().max()) < 5e-7
        add3 = torch.randn(*shape, 3, device="cuda", generator=gen)
        up3 = ops.lap_up(lo3, k, x3.shape, -5.0, addend=add3)
        up2 = ops.lap_up(lo2, k, x2.shape, -5.0, addend=add3[..., :2].contiguous())
        assert float((up3[..., :2] - up2).abs().max()) < 5e-6
        up1 = ops.lap_up(lo3[..., 2:3].contiguous(), k, shape + (1,), 5.0)    # one channel, no addend
        assert float((up1[..., 0] - ops.lap_up(lo3, k, x3.shape, 5.0)[..., 2]).abs().max()) < 5e-6


def test_laplacian_pyramid_normalisation_matches_the_oracle():
    """util.lap_normalize on the HIP kernels (strided 'SAME' smoothing, its transpose, RMS normalisation) vs the oracle's
    restatement with torch convolutions; even / odd sizes exercise both TF padding cases, c = 1 and 3"""
    from neural_flow_style_amd import util
    rng = np.random.RandomState(31)
    # (21 x 19 x 37 and 40 x 18 x 34: several LDS tiles of the 3-D kernels per axis, ragged last tiles, odd sizes)
    for shape, is_3d in (((12, 9, 10, 1), True), ((8, 8, 8, 3), True), ((21, 19, 37, 3), True), ((40, 18, 34, 1), True),
                         ((17, 12, 3), False), ((9, 9, 1), False)):
        g = rng.randn(*shape).astype(np.float32)
        k = util.lap_kernel(is_3d)
        for scale_n in (0, 1, 3) if min(shape[:-1]) >= 8 else (0, 1):
            want = O.lap_normalize(torch.tensor(g, dtype=torch.float64), k.astype(np.float64), scale_n).numpy()
            got = util.lap_normalize(torch.tensor(g).cuda(), scale_n=scale_n, is_3d=is_3d, c=shape[-1]).cpu().numpy()
            assert got.shape == g.shape
            assert rel(got, want) < 2e-5, (shape, scale_n)
    # volumes of >= 2^21 cells: the RMS normalisation of the high-pass levels rides in the kernels that write and merge
    # them (nfs_lap_up_rms) -- against the three-pass form of the same library and against the oracle
    import os
    import neural_flow_style_amd.ops as ops
    for shape in ((130, 128, 129, 3), (128, 132, 128, 1)):
        g = (rng.randn(*shape) * np.linspace(0.1, 30, shape[2])[None, None, :, None]).astype(np.float32)
        gd = torch.tensor(g).cuda()
        fused = util.lap_normalize(gd, scale_n=3, is_3d=True, c=shape[-1])
        os.environ["NFS_LAP_FUSE"] = "0"
        try:
            plain = util.lap_normalize(gd, scale_n=3, is_3d=True, c=shape[-1])
        finally:
            del os.environ["NFS_LAP_FUSE"]
        assert ops.lap_up_rms_parts(shape) > 0                                          # (the fused form exists for it)
        assert rel(fused.cpu(), plain.cpu()) < 1e-6
        assert torch.equal(fused, util.lap_normalize(gd, scale_n=3, is_3d=True, c=shape[-1]))      # deterministic
        if shape[-1] == 1:
            want = O.lap_normalize(torch.tensor(g, dtype=torch.float64), util.lap_kernel(True).astype(np.float64), 3).numpy()
            assert rel(fused.cpu().numpy(), want) < 2e-5
    # what it is for: every frequency band of the result has unit RMS before the merge, so a gradient dominated by one
    # scale is flattened -- the normalised field of a smooth + noisy mix has a higher noise-to-smooth ratio than the input
    zz, yy, xx = np.meshgrid(*[np.linspace(0, 1, 32)] * 3, indexing="ij")
    smooth = (100 * np.sin(2 * np.pi * zz) * np.cos(2 * np.pi * yy))[..., None].astype(np.float32)
    noise = rng.randn(32, 32, 32, 1).astype(np.float32)
    out = util.lap_normalize(torch.tensor(smooth + noise).cuda(), scale_n=3, is_3d=True).cpu().numpy()
    assert np.isfinite(out).all() and 0.5 < out.std() < 5
