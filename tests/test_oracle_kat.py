"""Oracle pinned to the reference's only known-answer vector
(transform.py:1859-1885) plus float64 gradchecks / closed forms."""
import numpy as np
import pytest
import torch

from oracle import nfs_oracle as O


def test_warp2d_docstring_kat():
    img = torch.arange(25, dtype=torch.float32).reshape(1, 5, 5, 1)
    ident = torch.tensor([1, 0, 0, 0, 1, 0], dtype=torch.float32)
    out = O.affine_warp2d(img, ident)[0, :, :, 0]
    assert torch.equal(out, img[0, :, :, 0])
    zoom = O.affine_warp2d(img, ident * 0.5)[0, :, :, 0].numpy()
    want = np.array([[6, 6.5, 7, 7.5, 8], [8.5, 9, 9.5, 10, 10.5], [11, 11.5, 12, 12.5, 13],
                     [13.5, 14, 14.5, 15, 15.5], [16, 16.5, 17, 17.5, 18]], np.float32)
    np.testing.assert_allclose(zoom, want, rtol=0, atol=1e-5)


def test_warp3d_reduces_to_kat_on_each_axis_pair():
    # embed the 5x5 KAT in a 3-D volume that is constant along one axis
    img2 = torch.arange(25, dtype=torch.float32).reshape(5, 5)
    want = O.affine_warp2d(img2.reshape(1, 5, 5, 1), torch.tensor([.5, 0, 0, 0, .5, 0]))[0, :, :, 0]
    vol = img2[None, :, :, None, None].expand(1, 5, 5, 4, 1).contiguous()
    g = O.mgrid(5, 5, 4).unsqueeze(0).clone()
    g[:, 0] *= 0.5; g[:, 1] *= 0.5
    out = O.batch_warp3d(vol, g, [1, 5, 5, 4])
    for k in range(4):
        assert torch.allclose(out[0, :, :, k, 0], want, atol=1e-5)


def test_interpolate3d_matches_grid_sample_border():
    torch.manual_seed(0)
    vol = torch.randn(1, 6, 7, 5, 2, dtype=torch.float64)
    c = torch.rand(1, 3, 6, 7, 5, dtype=torch.float64) * 3 - 1.5
    out = O.batch_warp3d(vol, c, [1, 6, 7, 5])
    grid = torch.stack([c[:, 2], c[:, 1], c[:, 0]], -1)
    ref = torch.nn.functional.grid_sample(vol.permute(0, 4, 1, 2, 3), grid, mode="bilinear",
                                          padding_mode="border", align_corners=True)
    assert torch.allclose(out, ref.permute(0, 2, 3, 4, 1), atol=1e-12)


def test_render_closed_form_adjoint():
    torch.manual_seed(1)
    d = torch.rand(2, 9, 4, 5, 1, dtype=torch.float64, requires_grad=True)
    g = torch.randn(2, 4, 5, 1, dtype=torch.float64)
    img = O.render_unnormalised(d, 0.3)
    (ga,) = torch.autograd.grad(img, d, g)
    gc = O.render_adjoint_closed_form(d.detach(), 0.3, g)
    assert torch.allclose(ga, gc, atol=1e-13)


@pytest.mark.parametrize("fn", ["advect", "rotate", "render", "render_liquid", "smooth"])
def test_gradcheck_ops(fn):
    torch.manual_seed(2)
    d = (torch.rand(1, 5, 6, 4, 1, dtype=torch.float64) + 0.1).requires_grad_()
    if fn == "advect":
        v = (torch.randn(1, 5, 6, 4, 3, dtype=torch.float64) * 0.2).requires_grad_()
        assert torch.autograd.gradcheck(O.advect, (d, v), eps=1e-6, atol=1e-6)
    elif fn == "rotate":
        R = torch.tensor(np.stack([np.eye(3), [[0.9, 0.1, 0], [-0.1, 0.9, 0.05], [0, -0.05, 1.0]]]))
        assert torch.autograd.gradcheck(lambda x: O.rotate(x, R), (d,), eps=1e-6, atol=1e-6)
    elif fn == "render":
        assert torch.autograd.gradcheck(lambda x: O.render(x, 0.2), (d,), eps=1e-6, atol=1e-6)
    elif fn == "render_liquid":
        assert torch.autograd.gradcheck(lambda x: O.render(x, 0.2, True), (d,), eps=1e-6, atol=1e-6)
    else:
        assert torch.autograd.gradcheck(lambda x: O.smooth3d_relu(x, 3), (d,), eps=1e-6, atol=1e-6)


def test_tf_maximum_gradient_passes_at_zero():
    x = torch.tensor([-1.0, 0.0, 2.0], requires_grad=True)
    O._TFMaximum0.apply(x).sum().backward()
    assert x.grad.tolist() == [0.0, 1.0, 1.0]


def test_p2g_vectorised_vs_loops():
    rng = np.random.RandomState(3)
    p = rng.uniform(-0.05, 1.05, (60, 3)).astype(np.float32)
    res = [8, 9, 7]; dom = [8., 9., 7.]
    a = O.p2g(torch.tensor(p, dtype=torch.float64)[None], dom, res, 0.5, 1000., 1, is_2d=False, clip=False).numpy()
    b = O.p2g_numpy_loops(p, dom, res, 0.5, 1000., 1, is_2d=False)
    np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-9)
    p2 = rng.uniform(0, 1, (40, 2)).astype(np.float32)
    a = O.p2g(torch.tensor(p2, dtype=torch.float64)[None], [3.2, 6.4], [16, 32], 0.025, 1000., 2, is_2d=True, clip=False).numpy()
    b = O.p2g_numpy_loops(p2, [3.2, 6.4], [16, 32], 0.025, 1000., 2, is_2d=True)
    np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-9)


def test_p2g_gradcheck_and_safe_sqrt():
    torch.manual_seed(4)
    p = torch.rand(1, 12, 3, dtype=torch.float64).requires_grad_()
    f = lambda q: O.p2g(q, [4., 4., 4.], [4, 4, 4], 0.5, 1000., 1, is_2d=False, clip=False)
    assert torch.autograd.gradcheck(f, (p,), eps=1e-7, atol=1e-5)
    # particle exactly on a cell centre: finite (zero) gradient instead of the reference's NaN
    pc = torch.tensor([[[0.375, 0.375, 0.375]]], dtype=torch.float64, requires_grad=True)
    f(pc).sum().backward()
    assert torch.isfinite(pc.grad).all()
    x = torch.rand(1, 12, 2, dtype=torch.float64).requires_grad_()
    fw = lambda q, a: O.p2g_wavg(q, a, [4., 4., 4.], [4, 4, 4], 0.5, 1, is_2d=False, clip=False)
    assert torch.autograd.gradcheck(fw, (p, x), eps=1e-7, atol=1e-5)


def test_tf_adam_differs_from_torch_adam_at_tiny_gradients():
    x0 = torch.ones(4); g = torch.full((4,), 1e-7)
    opt = O.TFAdam(); x = x0.clone()
    for _ in range(5):
        x = opt.step(x, g, 0.1)
    xt = x0.clone().requires_grad_()
    topt = torch.optim.Adam([xt], lr=0.1, eps=1e-8)
    for _ in range(5):
        xt.grad = g.clone(); topt.step()
    # closed form of TF ApplyAdam with constant gradient
    want = x0.clone(); m = 0.; v = 0.
    for t in range(1, 6):
        m = .9 * m + .1 * 1e-7; v = .999 * v + .001 * 1e-14
        want = want - 0.1 * np.sqrt(1 - .999 ** t) / (1 - .9 ** t) * m / (np.sqrt(v) + 1e-8)
    assert torch.allclose(x, want, rtol=1e-5)
    assert (x - xt.detach()).abs().max() > 0.05


def test_tf1_resize_bilinear_legacy_coords():
    x = torch.arange(4, dtype=torch.float32).reshape(1, 1, 4, 1)
    y = O.tf1_resize_bilinear(x, 1, 6)[0, 0, :, 0]
    # src = dst*4/6 -> 0, .667, 1.333, 2, 2.667, 3.333(clamped x1=3)
    np.testing.assert_allclose(y.numpy(), [0, 2 / 3, 4 / 3, 2, 8 / 3, 3], atol=1e-6)


def test_vgg_shapes_and_avgpool_valid():
    w = O.synthetic_vgg19_weights(width_div=16)
    f = O.vgg19_features(torch.rand(1, 25, 25, 3) * 255, w)
    assert f["conv1_1"].shape == (1, 25, 25, 4)
    assert f["conv2_1"].shape[1:3] == (12, 12)
    assert f["conv3_1"].shape[1:3] == (6, 6)
    assert f["conv5_1"].shape[1:3] == (1, 1)


def test_g2p_known_answers():
    """cell-centred sampling.  Linear g2p reproduces an affine field, cubic (Catmull-Rom)
    reproduces a quadratic, both exactly at cell centres; outside the centre lattice the border cell is
    replicated (both clipped cells coincide, transform.py:1145-1154)."""
    torch.manual_seed(0)
    X, Y, Z = 8, 7, 9
    zz, yy, xx = torch.meshgrid(torch.arange(X) + 0.5, torch.arange(Y) + 0.5, torch.arange(Z) + 0.5, indexing="ij")
    aff = (2 * zz - 3 * yy + 0.5 * xx + 1).double()[None, ..., None]
    p = torch.rand(1, 64, 3).double() * 0.4 + 0.3                      # well inside: no clipped stencil cell
    want = 2 * p[0, :, 0] * X - 3 * p[0, :, 1] * Y + 0.5 * p[0, :, 2] * Z + 1
    for lin in (True, False):
        got = O.g2p(aff, p, is_2d=False, is_linear=lin)[0, :, 0]
        assert float((got - want).abs().max()) < 1e-12
    quad = (zz ** 2 - yy * xx).double()[None, ..., None]
    got = O.g2p(quad, p, is_2d=False)[0, :, 0]
    want = (p[0, :, 0] * X) ** 2 - (p[0, :, 1] * Y) * (p[0, :, 2] * Z)
    assert float((got - want).abs().max()) < 1e-12
    # at a cell centre both interpolants return the cell value
    g = torch.randn(1, 5, 6, 2, dtype=torch.float64)
    pc = torch.tensor([[[2.5 / 5, 3.5 / 6]]], dtype=torch.float64)
    for lin in (True, False):
        assert torch.allclose(O.g2p(g, pc, is_2d=True, is_linear=lin)[0, 0], g[0, 2, 3])
    # outside [0.5, n-0.5] the linear form replicates the border cell
    po = torch.tensor([[[-0.2, 3.5 / 6], [1.3, 3.5 / 6], [0.01, 0.99]]], dtype=torch.float64)
    out = O.g2p(g, po, is_2d=True, is_linear=True)[0]
    assert torch.allclose(out[0], g[0, 0, 3]) and torch.allclose(out[1], g[0, 4, 3])
    assert torch.allclose(out[2], g[0, 0, 5])


def test_mac_to_centered_and_rk4_advect():
    # a uniform MAC field stays uniform in the interior after face averaging (the zero-padded high face halves it)
    v = np.ones((4, 5, 6, 3), np.float32) * np.array([1.0, 2.0, 3.0], np.float32)
    c = O.mac_to_centered(v)
    assert c.shape == v.shape
    np.testing.assert_allclose(c[:-1, 1:, :-1], np.broadcast_to([1.0, 2.0, 3.0], (3, 4, 5, 3)))
    np.testing.assert_allclose(c[0, 0, -1, 0], 0.5)
    # constant velocity field: RK4 = explicit Euler, x_adv = x + 0.5 u
    u = torch.zeros(6, 6, 6, 3, dtype=torch.float64) + torch.tensor([0.01, -0.02, 0.03], dtype=torch.float64)
    x = torch.rand(10, 3, dtype=torch.float64) * 0.5 + 0.25
    assert torch.allclose(O.simg2p_advect(x, u), x + 0.5 * u[0, 0, 0])


def test_content_loss_known_answers():
    """styler_base.py:135-150 on a 1x1x2x4 feature, by hand"""
    f = torch.tensor([[[[1.0, 2.0, 3.0, 4.0], [0.0, 6.0, 1.0, 2.0]]]])
    # channel 2: -mean([3,1]) + mean|[1,2,0,6]| + mean|[4,2]| = -2 + 2.25 + 3
    assert abs(float(O.content_loss(f, 2)) - 3.25) < 1e-6
    # channel falsy: -mean(all) = -19/8
    assert abs(float(O.content_loss(f, 0)) + 19.0 / 8.0) < 1e-6
    # content image: mean((f - 2*t)^2) with t = f/2 + 1/2 -> (f - f - 1)^2 = 1
    assert abs(float(O.content_loss(f, 0, f / 2 + 0.5, 2.0)) - 1.0) < 1e-6
    # the last channel has no upper slice
    assert abs(float(O.content_loss(f, 3)) - (-3.0 + 13.0 / 6.0)) < 1e-6


def test_histogram_match_known_answers_and_the_skipped_cases():
    """util.histogram_match_tf (util.py:317-399) by hand on a case small enough to follow, plus the two cases the
    reference leaves undefined and this build defines as "channel skipped" (matched = source: loss 0, gradient 0)"""
    # source uniform on [0, 254] in steps of 1, template = the same values doubled in count: identical CDFs on the joint
    # range [0, 254] -> every source value maps to the centre of its own bin (delta = 254/255)
    src = np.arange(255, dtype=np.float32)
    tpl = np.repeat(src, 2)
    m = O.histogram_match(src, tpl)
    delta = np.float32(254.0 / 255.0)
    k = np.clip((src / delta).astype(np.int64), 0, 254)
    # identical CDFs: nearest_indices is the identity up to the round() of the interpolated bin index
    assert np.abs(m - ((delta * k).astype(np.float32) + delta / 2)).max() <= 1.01 * delta
    # a template living in the upper half of the range pulls every source value there
    src = np.linspace(0, 1, 101, dtype=np.float32)
    tpl = np.linspace(3, 4, 50, dtype=np.float32)
    m = O.histogram_match(src, tpl)
    assert m.min() >= 3 - 4 / 255 and m.max() <= 4 + 4 / 255 and np.all(np.diff(m) >= 0)
    # flat channel and empty source: skipped
    f = np.full((4, 3), 2.5, np.float32)
    assert np.array_equal(O.histogram_match(f, np.full((5,), 2.5, np.float32)), f)
    assert O.histogram_match(np.zeros((0,), np.float32), tpl).size == 0
    # masked loss: only the pixels under a non-zero mask enter the match and the sum
    rng = np.random.RandomState(0)
    feat = torch.tensor(rng.rand(1, 6, 5, 2).astype(np.float32) * 10, requires_grad=True)
    templ = torch.tensor(rng.rand(1, 4, 4, 2).astype(np.float32) * 10)
    mask = (rng.rand(1, 6, 5, 1) < 0.5).astype(np.float32)
    l = O.hist_loss(feat, templ, mask=torch.tensor(mask))
    (g,) = torch.autograd.grad(l, feat)
    assert float(g[0][torch.tensor(mask[0, ..., 0]) == 0].abs().max()) == 0.0
    sel = mask[0, ..., 0] != 0
    want = sum(float(((feat.detach().numpy()[0, ..., j][sel]
                       - O.histogram_match(feat.detach().numpy()[0, ..., j][sel], templ.numpy()[0, ..., j])) ** 2).sum())
               for j in range(2))
    assert abs(float(l.detach()) - want) < 1e-4 * want
    assert float(O.hist_loss(feat, templ, mask=torch.zeros(1, 6, 5, 1))) == 0.0
