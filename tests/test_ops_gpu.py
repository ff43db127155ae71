"""Operator parity: every HIP entry point vs the CPU oracle on the same seeded
inputs (through the C ABI).  Tolerance: relative L2 <= 1e-4 per operator for
float32 (north_star's end-to-end bar is 1e-3; float atomics make the scatter
adjoints run-to-run non-deterministic at ~1e-7)."""
import numpy as np
import pytest
import torch

from oracle import nfs_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4


def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def ops():
    import neural_flow_style_amd.ops as ops
    return ops


def dev(x):
    return x.detach().float().contiguous().cuda()


def rots(n, seed=0, big=False):
    import neural_flow_style_amd.transform as T
    rng = np.random.RandomState(seed)
    m = [np.eye(3)]
    for _ in range(n - 1):
        phi, th = rng.uniform(-5, 5), rng.uniform(-10, 10)
        if big:
            phi, th = rng.uniform(-40, 40), rng.uniform(-60, 60)
        m.append(T.rot_y_3d(th) @ T.rot_z_3d(phi))
    return torch.tensor(np.stack(m), dtype=torch.float32)


@pytest.mark.parametrize("shape", [(9, 7, 11, 1), (6, 8, 5, 3)])
def test_warp3d(ops, shape):
    torch.manual_seed(0)
    X, Y, Z, C = shape
    imgs = torch.randn(2, X, Y, Z, C)
    coords = torch.rand(2, 3, X, Y, Z) * 2.6 - 1.3
    imgs_r = imgs.clone().requires_grad_(); coords_r = coords.clone().requires_grad_()
    ref = O.batch_warp3d(imgs_r, coords_r, [2, X, Y, Z])
    out = ops.warp3d_fwd(dev(imgs), dev(coords))
    assert rel(out, ref) < TOL
    g = torch.randn_like(ref)
    gi, gc = torch.autograd.grad(ref, (imgs_r, coords_r), g)
    gi_h, gc_h = ops.warp3d_bwd(dev(imgs), dev(coords), dev(g))
    assert rel(gi_h, gi) < TOL
    assert rel(gc_h, gc) < TOL


@pytest.mark.parametrize("big", [False, True])
def test_rotate(ops, big):
    torch.manual_seed(1)
    d = torch.rand(1, 10, 12, 9, 2).requires_grad_()
    R = rots(3, 1, big)
    ref = O.rotate(d, R)
    out = ops.rotate_fwd(dev(d[0]), dev(R))
    assert rel(out, ref) < TOL
    g = torch.randn_like(ref)
    (gd,) = torch.autograd.grad(ref, d, g)
    gd_h = ops.rotate_bwd(dev(g), dev(R))
    assert rel(gd_h, gd[0]) < TOL


@pytest.mark.parametrize("shape,big", [((10, 12, 9), False), ((21, 37, 45), True), ((40, 33, 70), False),
                                       ((1, 20, 24), True)])
def test_rotate_bwd_tiled_c1(ops, shape, big):
    """C = 1 takes the LDS-tiled output-stationary adjoint (no global atomics); it must equal the
    autograd adjoint for small and large rotations, non-cubic volumes and clamped (out-of-cube) samples."""
    torch.manual_seed(21)
    D, H, W = shape
    d = torch.rand(1, D, H, W, 1).requires_grad_()
    R = rots(4, 3, big)
    R[3] = R[3] * 1.4          # not a pure rotation: scaling pushes many samples outside the cube
    ref = O.rotate(d, R)
    g = torch.randn_like(ref)
    (gd,) = torch.autograd.grad(ref, d, g)
    gd_h = ops.rotate_bwd(dev(g), dev(R))
    assert rel(gd_h, gd[0]) < TOL
    assert torch.equal(gd_h, ops.rotate_bwd(dev(g), dev(R)))      # fixed-point accumulation: bit-reproducible
    assert rel(ops.rotate_bwd(dev(g), dev(R), tiled=False), gd[0]) < TOL   # global-atomic fallback
    for s in (1e-20, 1e20):                                        # the fixed-point scale follows max|g|
        assert rel(ops.rotate_bwd(dev(g * s), dev(R)), gd[0] * s) < TOL
    assert float(ops.rotate_bwd(dev(g * 0), dev(R)).abs().max()) == 0.0
    # accumulate semantics: += into a pre-filled buffer
    acc = torch.ones(D, H, W, 1, device="cuda")
    ops.rotate_bwd(dev(g), dev(R), g_d_acc=acc)
    assert rel(acc - 1.0, gd[0]) < TOL


@pytest.mark.parametrize("shape", [(11, 9, 13), (12, 10, 16), (8, 6, 2)])
def test_advect(ops, shape):
    """(11,9,13): generic one-voxel kernels; n % 4 == 0 shapes: the 4-voxel scalar kernels"""
    torch.manual_seed(2)
    D, H, W = shape
    d = torch.rand(1, D, H, W, 1).requires_grad_()
    v = (torch.randn(1, D, H, W, 3) * 0.25).requires_grad_()
    ref = O.advect(d, v)
    out = ops.advect_fwd(dev(d[0]), dev(v[0]))
    assert rel(out, ref[0]) < TOL
    g = torch.randn_like(ref)
    gd, gv = torch.autograd.grad(ref, (d, v), g)
    gd_h, gv_h = ops.advect_bwd(dev(d[0]), dev(v[0]), dev(g[0]))
    assert rel(gd_h, gd[0]) < TOL
    assert rel(gv_h, gv[0]) < TOL
    _, gv_only = ops.advect_bwd(dev(d[0]), dev(v[0]), dev(g[0]), need_d=False)     # velocity-only adjoint
    assert rel(gv_only, gv[0]) < TOL
    d2 = torch.rand(1, D, H, W, 2)                                                  # C > 1 stays generic
    assert rel(ops.advect_fwd(dev(d2[0]), dev(v[0])), O.advect(d2, v.detach())[0]) < TOL


@pytest.mark.parametrize("k", [3.0, 0.0])
@pytest.mark.parametrize("shape", [(7, 9, 13), (8, 8, 16), (29, 19, 131)])   # last: 2 z-chunks, 3 row tiles, 3 column tiles
def test_smooth3d_relu(ops, k, shape):
    torch.manual_seed(3)
    d = torch.randn(1, *shape, 1)
    d[0, :2] = 0.0  # exact zeros: TF's Maximum gradient passes at pre == 0
    d = d.requires_grad_()
    ref = O.smooth3d_relu(d, k)
    out = ops.smooth3d_relu_fwd(dev(d[0, ..., 0]), k)
    assert rel(out, ref[0, ..., 0]) < TOL
    g = torch.randn_like(ref)
    (gd,) = torch.autograd.grad(ref, d, g)
    gd_h = ops.smooth3d_relu_bwd(out, dev(g[0, ..., 0]), k)
    assert rel(gd_h, gd[0, ..., 0]) < TOL


@pytest.mark.parametrize("shape,V,big", [((40, 37, 50), 3, False), ((64, 20, 33), 1, True), ((17, 16, 16), 2, False),
                                         ((33, 5, 9), 4, True)])
def test_rotate_render_adjoint_coefficient_form(ops, shape, V, big):
    """the u / coefficient form of the rotate + render adjoint (no render-adjoint pass over the rotated volume) against
    the oracle and against the two-pass form: same image and ray sums bit for bit, gradient to float32 rounding"""
    torch.manual_seed(11)
    D, H, W = shape
    tau = 0.07
    d = torch.rand(1, D, H, W, 1).requires_grad_()
    R = rots(V + 1, 7, big)[1:]                         # (no identity view)
    rot = dev(R)
    assert ops.render_coef_layout(V, D, H, W) is not None
    dd = dev(d[0, ..., 0])
    img, rs, u_rot, seg = ops.rotate_render_fwd_coef(dd, rot, tau)
    d_rot = torch.empty(V, D, H, W, device="cuda")
    img2, rs2 = ops.rotate_render_fwd(dd, rot, tau, 0, d_rot=d_rot)
    assert torch.equal(img, img2) and torch.equal(rs, rs2)
    g_img = torch.randn(V, H, W, device="cuda")
    g_img[0, :2] = 0.0                                   # rays without gradient: skipped samples
    ab, gmax = ops.render_ray_coef(g_img, seg, tau)
    g_new = ops.rotate_bwd_coef(u_rot, ab, rot, gmax)
    g_rot, gm2 = ops.render_bwd(d_rot, rs2, g_img, tau, 0, want_max=True)
    assert float(gmax.max()) >= float(gm2) * (1 - 1e-6)  # (per-block bounds) a bound on every sample gradient
    g_old = ops.rotate_bwd(g_rot.unsqueeze(-1), rot, g_max=gm2)[..., 0]
    assert rel(g_new, g_old.cpu()) < 2e-6
    acc = torch.ones(D, H, W, device="cuda")             # accumulate form
    ops.rotate_bwd_coef(u_rot, ab, rot, gmax, g_d_acc=acc)
    assert rel(acc - 1.0, g_old.cpu()) < 1e-5
    # oracle: autograd through rotate + the un-normalised transmittance integral
    img_o = O.render_unnormalised(O.rotate(d, R), tau)
    assert rel(img, img_o[..., 0]) < TOL
    (g_o,) = torch.autograd.grad(img_o, d, g_img.cpu().unsqueeze(-1))
    assert rel(g_new, g_o[0, ..., 0]) < TOL
    # a shape the segmented forward does not take: the caller is told to keep the two-pass form
    assert ops.render_coef_layout(1, 8, 8, 8) is None


def test_smooth3d_relu_sixteen_row_tiles(ops):
    """from 256 blocks on the kernel runs two rows per thread (16-row tiles, column tiles <= 54): a ragged volume large
    enough to take that instance (4 x 9 x 8 blocks; edge tiles in every direction) against the oracle, and bit for bit
    against the same planes computed as part of a thin slab that still takes the 8-row instance"""
    torch.manual_seed(5)
    shape = (190, 139, 213)
    d = torch.randn(1, *shape, 1)
    d[0, :2] = 0.0
    d = d.requires_grad_()
    ref = O.smooth3d_relu(d, 3.0)
    out = ops.smooth3d_relu_fwd(dev(d[0, ..., 0]), 3.0)
    assert rel(out, ref[0, ..., 0]) < TOL
    g = torch.randn_like(ref)
    (gd,) = torch.autograd.grad(ref, d, g)
    gd_h = ops.smooth3d_relu_bwd(out, dev(g[0, ..., 0]), 3.0)
    assert rel(gd_h, gd[0, ..., 0]) < TOL
    # planes 40 .. 59 as the interior of a 22-plane slab (2 z-chunks x 3 x 9 blocks < 256: the 8-row instance)
    slab = ops.smooth3d_relu_fwd(dev(d[0, 39:61, :, :, 0].detach().contiguous()), 3.0)
    assert torch.equal(slab[1:-1], out[40:60])
    gslab = ops.smooth3d_relu_bwd(out[38:62].contiguous(), dev(g[0, 38:62, :, :, 0].contiguous()), 3.0)
    assert torch.equal(gslab[1:-1], gd_h[39:61])


@pytest.mark.parametrize("liquid", [False, True])
def test_render_and_maxnorm(ops, liquid):
    torch.manual_seed(4)
    d = torch.rand(3, 17, 8, 9, 1).requires_grad_()
    tau = 0.2
    ref = O.render(d, tau, liquid)  # whole batch = one normalisation group (v_batch = 3)
    img, rs = ops.render_fwd(dev(d[..., 0]), tau, liquid)
    if liquid:
        out = img
    else:
        out, gmax = ops.maxnorm_fwd(img, 1)
    assert rel(out, ref[..., 0]) < TOL
    g = torch.randn_like(ref)
    (gd,) = torch.autograd.grad(ref, d, g)
    gi = dev(g[..., 0])
    if not liquid:
        gi = ops.maxnorm_bwd(img, gmax, gi)
    gd_h = ops.render_bwd(dev(d[..., 0]), rs, gi, tau, liquid)
    assert rel(gd_h, gd[..., 0]) < TOL


@pytest.mark.parametrize("sign", [1.0, -1.0])
def test_maxnorm_multiblock(ops, sign):
    """groups of >= 16384 elements take the 32-block two-phase path: float max through integer atomics (also for
    all-negative groups), deterministic partial sums, ties split like TF"""
    torch.manual_seed(12)
    x = (torch.rand(3, 150, 140) + 0.1) * sign
    x[1, 7, 9] = x[1, 100, 3] = x[1].max() + 0.5 if sign > 0 else x[1].max() * 0.5    # a tie at the maximum
    x = x.requires_grad_()
    y = x / x.amax(dim=(1, 2), keepdim=True)
    g = torch.randn_like(x)
    (gx,) = torch.autograd.grad(y, x, g)
    out, gmax = ops.maxnorm_fwd(dev(x), 3)
    assert torch.equal(gmax.cpu(), x.detach().amax(dim=(1, 2))) and rel(out, y) < 1e-6
    gx_h = ops.maxnorm_bwd(dev(x), gmax, dev(g))
    assert rel(gx_h, gx) < 1e-5
    assert torch.equal(gx_h, ops.maxnorm_bwd(dev(x), gmax, dev(g)))


def test_maxnorm_ties_split_like_tf(ops):
    x = torch.tensor([[1.0, 3.0, 3.0, 2.0]]).requires_grad_()
    y = x / x.amax()
    g = torch.tensor([[0.5, -1.0, 2.0, 0.25]])
    (gx,) = torch.autograd.grad(y, x, g)
    out, gmax = ops.maxnorm_fwd(dev(x), 1)
    gx_h = ops.maxnorm_bwd(dev(x), gmax, dev(g))
    assert rel(gx_h, gx) < 1e-6


@pytest.mark.parametrize("D", [14, 21, 32])   # 14: one thread per ray; >= 16: 4 depth segments per ray (21: ragged)
@pytest.mark.parametrize("liquid", [False, True])
def test_rotate_render_fused(ops, liquid, D):
    torch.manual_seed(5)
    d = torch.rand(1, D, 12, 10, 1).requires_grad_()
    R = rots(4, 5)
    tau = 0.15
    dr = O.rotate(d, R)
    ref = torch.cat([O.render(dr[v:v + 1], tau, liquid) for v in range(4)])  # per-view max (v_batch=1)
    img, rs = ops.rotate_render_fwd(dev(d[0, ..., 0]), dev(R), tau, liquid)
    if liquid:
        out = img
    else:
        out, gmax = ops.maxnorm_fwd(img, 4)
    assert rel(out, ref[..., 0]) < TOL
    g = torch.randn_like(ref)
    (gd,) = torch.autograd.grad(ref, d, g)
    gi = dev(g[..., 0])
    if not liquid:
        gi = ops.maxnorm_bwd(img, gmax, gi)
    gd_h = ops.rotate_render_bwd(dev(d[0, ..., 0]), dev(R), rs, gi, tau, liquid)
    assert rel(gd_h, gd[0, ..., 0]) < TOL
    # two-pass adjoint: kept rotated volume -> in-place render adjoint -> tiled rotate adjoint
    d_rot = torch.empty(4, D, 12, 10, device="cuda")
    img3, rs3 = ops.rotate_render_fwd(dev(d[0, ..., 0]), dev(R), tau, liquid, d_rot=d_rot)
    assert rel(img3, img) < 1e-6 and rel(d_rot, dr[..., 0]) < TOL
    g_rot, g_max = ops.render_bwd(d_rot, rs3, gi, tau, liquid, g_d=d_rot, want_max=True)
    assert float(g_max) == float(g_rot.abs().max())                   # by-product of the same pass, exact
    gd_2 = ops.rotate_bwd(g_rot.unsqueeze(-1), dev(R))
    assert rel(gd_2[..., 0], gd[0, ..., 0]) < TOL
    gd_3 = ops.rotate_bwd(g_rot.unsqueeze(-1), dev(R), g_max=g_max)    # supplied maximum: no pre-pass
    assert torch.equal(gd_3, gd_2)
    # fused == unfused HIP
    img2, _ = ops.render_fwd(ops.rotate_fwd(dev(d[0]), dev(R))[..., 0].contiguous(), tau, liquid)
    assert rel(img2, img) < 1e-5


@pytest.mark.parametrize("cin,scale", [(1, 1.0), (3, 1.0), (1, 1.5)])
def test_loss_net_input(ops, cin, scale):
    torch.manual_seed(6)
    img = torch.rand(2, 10, 12, cin).requires_grad_()
    ref = O.plugin_to_loss_net(img, scale, is_color=(cin == 3))
    H2, W2 = ref.shape[1:3]
    d_img, x = ops.loss_net_input_fwd(dev(img), H2, W2)
    assert rel(d_img, ref) < TOL
    assert rel(x, ref - torch.tensor(O.VGG_MEAN)) < TOL
    g = torch.randn_like(ref)
    (gi,) = torch.autograd.grad(ref, img, g)
    gi_h = ops.loss_net_input_bwd(dev(g), 10, 12, cin)
    assert rel(gi_h, gi) < TOL


def _conv_ref(x, w, b, relu=True):
    y = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), torch.as_tensor(w).permute(3, 2, 0, 1),
                                   None if b is None else torch.as_tensor(b), padding=1)
    if relu:
        y = torch.relu(y)
    return y.permute(0, 2, 3, 1)


@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 13, 11, 3, 64), (2, 13, 11, 64, 64), (3, 25, 25, 64, 128),
                                          (1, 12, 12, 128, 256), (2, 7, 50, 128, 128), (8, 6, 6, 256, 128),
                                          (1, 25, 25, 512, 512), (1, 12, 12, 512, 512),    # these two: K-split GEMMs
                                          # the single-kernel Winograd path (64 / 128 channels, whole 4x4 tiles): runs of
                                          # 16 tiles that wrap rows and images, a ragged last run, both channel groups
                                          # (with 128 input channels it is taken from 128 blocks of 16 tiles x 64 channels on)
                                          (3, 20, 28, 64, 64), (2, 12, 16, 128, 128), (5, 8, 8, 128, 64),
                                          (1, 4, 4, 64, 128), (4, 64, 64, 128, 128), (8, 64, 64, 128, 64),
                                          # ... and its ragged instance (H or W off a multiple of 4: the last tile row /
                                          # column hangs over the image; per-pixel validity in the epilogue)
                                          (2, 30, 45, 64, 64), (2, 30, 45, 64, 128), (4, 62, 66, 128, 128),
                                          (4, 61, 67, 128, 64), (1, 5, 6, 64, 64)])
def test_conv3x3_fwd_and_dgrad(ops, B, H, W, Ci, Co):
    rng = np.random.RandomState(7)
    x = torch.tensor(rng.randn(B, H, W, Ci), dtype=torch.float32).requires_grad_()
    w = (rng.randn(3, 3, Ci, Co) * np.sqrt(2.0 / (9 * Ci))).astype(np.float32)
    b = (rng.randn(Co) * 0.1).astype(np.float32)
    ref = _conv_ref(x, w, b)
    wf = ops.conv3x3_pack(dev(torch.tensor(w)), 0)
    out = ops.conv3x3_fwd(dev(x), wf, dev(torch.tensor(b)), Co, relu=True)
    assert rel(out, ref) < TOL
    # data gradient wrt x given the gradient wrt the pre-activation
    gy = torch.tensor(rng.randn(B, H, W, Co), dtype=torch.float32)
    pre = _conv_ref(x, w, b, relu=False)
    (gx,) = torch.autograd.grad(pre, x, gy)
    wd = ops.conv3x3_pack(dev(torch.tensor(w)), 1)
    gx_h = ops.conv3x3_dgrad(dev(gy), wd, Ci)
    assert rel(gx_h, gx) < TOL
    if Ci != 3:
        xin = torch.tensor(rng.randn(B, H, W, Ci), dtype=torch.float32)
        add = torch.tensor(rng.randn(B, H, W, Ci), dtype=torch.float32)
        gx_m = ops.conv3x3_dgrad(dev(gy), wd, Ci, x_in=dev(xin), addend=dev(add))
        assert rel(gx_m, gx * (xin > 0) + add) < TOL


@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 25, 25, 512, 512), (1, 50, 50, 256, 256), (1, 23, 48, 128, 256)])
def test_winograd_f5_error_budget(ops, B, H, W, Ci, Co):
    """the deep layers with heavily padded F(4x4) tilings run F(5x5,3x3) (interpolation points 0, +-1, +-2, 1/2, inf) in
    float32.  Stated budget against a float64 convolution, at activations of O(10^2..10^3) as the real vgg_19 produces
    them: 2e-5 relative L2 for the forward pass and for the data gradient (measured ~5e-6; F(4x4) ~2.5e-6, the direct
    form ~7e-7)"""
    rng = np.random.RandomState(23)
    x = torch.tensor(np.maximum(rng.randn(B, H, W, Ci), 0) * 300.0, dtype=torch.float32)
    w = (rng.randn(3, 3, Ci, Co) * np.sqrt(2.0 / (9 * Ci))).astype(np.float32)
    b = (rng.randn(Co) * 1.0).astype(np.float32)
    x64 = x.double().requires_grad_()
    w64 = torch.tensor(w).double().permute(3, 2, 0, 1)
    pre = torch.nn.functional.conv2d(x64.permute(0, 3, 1, 2), w64, torch.tensor(b).double(), padding=1).permute(0, 2, 3, 1)
    wf = ops.conv3x3_pack(dev(torch.tensor(w)), 0)
    out = ops.conv3x3_fwd(dev(x), wf, dev(torch.tensor(b)), Co, relu=True)
    assert rel(out, torch.relu(pre).detach()) < 2e-5
    gy = torch.tensor(rng.randn(B, H, W, Co), dtype=torch.float32)
    (gx,) = torch.autograd.grad(pre, x64, gy.double())
    wd = ops.conv3x3_pack(dev(torch.tensor(w)), 1)
    assert rel(ops.conv3x3_dgrad(dev(gy), wd, Ci), gx) < 2e-5


@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 13, 11, 64, 64), (1, 25, 25, 128, 128), (2, 8, 12, 64, 128),
                                          (3, 20, 28, 64, 64), (2, 12, 16, 128, 128), (5, 8, 8, 128, 64),
                                          (4, 64, 64, 128, 128),
                                          (1, 24, 24, 512, 512),      # 36 tiles, K = 512: the GEMM runs in two K parts
                                          # ragged single-kernel instances: odd sides floor in the pool (a window exists
                                          # when its lower right pixel does), the pooled gradient is zero beyond them
                                          (2, 30, 45, 64, 64), (4, 61, 66, 128, 128), (2, 150, 225, 128, 128)])
def test_conv3x3_fused_pool(ops, B, H, W, Ci, Co):
    """conv + ReLU + 2x2 VALID average pool in one pass, and the data gradient taken from the POOLED gradient
    (pool adjoint + ReLU mask folded into the Winograd input transform); odd sizes floor like slim.avg_pool2d"""
    rng = np.random.RandomState(17)
    x = torch.tensor(rng.randn(B, H, W, Ci), dtype=torch.float32).requires_grad_()
    w = (rng.randn(3, 3, Ci, Co) * np.sqrt(2.0 / (9 * Ci))).astype(np.float32)
    b = (rng.randn(Co) * 0.1).astype(np.float32)
    y = _conv_ref(x, w, b)
    yp = torch.nn.functional.avg_pool2d(y.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    wf = ops.conv3x3_pack(dev(torch.tensor(w)), 0)
    y_h, yp_h = ops.conv3x3_fwd_pool(dev(x), wf, dev(torch.tensor(b)), Co, relu=True)
    assert rel(y_h, y) < TOL and rel(yp_h, yp) < TOL
    g = torch.tensor(rng.randn(*yp.shape), dtype=torch.float32)
    # the gradient through pool and ReLU, with the ReLU mask taken from the kernel's own output: at large sizes a few
    # of the millions of pre-activations sit within rounding of zero, and a mask that flips there moves the gradient
    # by far more than any rounding error (autograd through the reference's own ReLU is used at the small sizes)
    pre = _conv_ref(x, w, b, relu=False)
    gpre = torch.zeros_like(pre)
    PH, PW = H // 2, W // 2
    up = 0.25 * g.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    gpre[:, :2 * PH, :2 * PW] = up * (y_h.cpu()[:, :2 * PH, :2 * PW] > 0)
    (gx,) = torch.autograd.grad(pre, x, gpre, retain_graph=True)
    wd = ops.conv3x3_pack(dev(torch.tensor(w)), 1)
    gx_h = ops.conv3x3_dgrad_pool(dev(g), y_h, wd, Ci)
    assert rel(gx_h, gx) < TOL
    if B * H * W * Co < 200000:
        (gx_auto,) = torch.autograd.grad(yp, x, g)
        assert rel(gx_h, gx_auto) < TOL
    xin = torch.tensor(rng.randn(B, H, W, Ci), dtype=torch.float32)
    add = torch.tensor(rng.randn(B, H, W, Ci), dtype=torch.float32)
    gx_m = ops.conv3x3_dgrad_pool(dev(g), y_h, wd, Ci, x_in=dev(xin), addend=dev(add))
    assert rel(gx_m, gx * (xin > 0) + add) < TOL


def test_avgpool2(ops):
    torch.manual_seed(8)
    x = torch.randn(2, 7, 9, 8).requires_grad_()
    ref = torch.nn.functional.avg_pool2d(x.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    out = ops.avgpool2_fwd(dev(x))
    assert rel(out, ref) < TOL
    g = torch.randn_like(ref)
    (gx,) = torch.autograd.grad(ref, x, g)
    gx_h = ops.avgpool2_bwd(dev(g), x.shape)
    assert rel(gx_h, gx) < TOL
    add = torch.randn_like(x)
    gx_m = ops.avgpool2_bwd(dev(g), x.shape, x=dev(x), addend=dev(add))
    assert rel(gx_m, gx * (x > 0) + add) < TOL


@pytest.mark.parametrize("B,HW,C", [(2, 300, 64), (3, 5003, 64), (1, 1000, 128), (3, 37, 256), (1, 144, 512),
                                    (8, 144, 512)])     # last: >= 256 tile pairs and a short image -> one slab, no reduce pass;
                                                        # C = 64: the streaming Gram gradient (ragged last row tile, several runs)
def test_gram_style_loss(ops, B, HW, C):
    torch.manual_seed(9)
    F = torch.relu(torch.randn(B, HW, C)).requires_grad_()
    Fs = torch.relu(torch.randn(1, HW, C))
    denom = 2.0 * HW * C
    G = torch.stack([f.t() @ f for f in F]) / denom
    Gs = torch.stack([f.t() @ f for f in Fs]) / denom
    wgt = 0.7
    loss = wgt * ((G - Gs) ** 2).sum()
    (gF,) = torch.autograd.grad(loss, F)
    Gh = ops.gram_fwd(dev(F), 1.0 / denom)                          # two-pass deterministic reduction
    Gsh = ops.gram_fwd(dev(Fs), 1.0 / denom)
    assert rel(Gh, G) < TOL
    assert torch.equal(Gh, ops.gram_fwd(dev(F), 1.0 / denom))
    assert rel(ops.gram_fwd(dev(F), 1.0 / denom, two_pass=False), G) < TOL   # float-atomic fallback
    lacc = torch.zeros(B, device="cuda")
    Dm = ops.style_loss_fwd(Gh, Gsh, wgt, lacc)
    assert abs(float(lacc.sum()) - float(loss)) / float(loss) < TOL
    dF = ops.gram_bwd(dev(F), Dm, 1.0 / denom, relu_mask=False)
    assert rel(dF, gF) < TOL
    dFm = ops.gram_bwd(dev(F), Dm, 1.0 / denom, relu_mask=True)
    assert rel(dFm, gF * (F > 0)) < TOL


def test_tv_loss(ops):
    torch.manual_seed(10)
    x = (torch.rand(2, 9, 11, 3) * 255).requires_grad_()
    ref = O.tv_loss(x) * 0.01
    (gx,) = torch.autograd.grad(ref, x)
    lacc = torch.zeros(1, device="cuda"); g = torch.zeros(x.shape, device="cuda")
    ops.tv_loss(dev(x), 0.01, lacc, g)
    assert abs(float(lacc) - float(ref)) / float(ref) < TOL
    assert rel(g, gx) < TOL


def test_adam_tf(ops):
    torch.manual_seed(11)
    x0 = torch.randn(1003); opt = O.TFAdam()
    x = dev(x0); m = torch.zeros_like(x); v = torch.zeros_like(x)
    xr = x0.clone()
    b1p = np.float32(1); b2p = np.float32(1)
    for t in range(5):
        g = torch.randn(1003) * (1e-7 if t % 2 else 1.0)
        xr = opt.step(xr, g, 0.1)
        b1p = np.float32(b1p * np.float32(0.9)); b2p = np.float32(b2p * np.float32(0.999))
        lr_t = np.float32(0.1) * np.sqrt(np.float32(1) - b2p) / (np.float32(1) - b1p)
        ops.adam_tf_step(x, m, v, dev(g), float(lr_t))
    assert rel(x, xr) < 1e-5


@pytest.mark.parametrize("nd", [2, 3])
def test_p2g_density(ops, nd):
    rng = np.random.RandomState(12)
    N = 500
    p = torch.tensor(rng.uniform(-0.03, 1.03, (1, N, nd)), dtype=torch.float32).requires_grad_()
    if nd == 3:
        res, dom, radius, nsize = [12, 14, 10], [12., 14., 10.], 0.5, 1
    else:
        res, dom, radius, nsize = [16, 32], [1.6, 3.2], 0.025, 2
    ref = O.p2g(p, dom, res, radius, 1000., nsize, is_2d=(nd == 2), clip=False)
    cfg = ops.make_splat_cfg(nd, res, dom, radius, 4, 1000., nsize, False, 0)
    out = ops.p2g_fwd(dev(p[0]), cfg)
    assert rel(out, ref[0]) < TOL
    g = torch.randn_like(ref)
    (gp,) = torch.autograd.grad(ref, p, g)
    gp_h, _, _ = ops.p2g_bwd(dev(p[0]), cfg, dev(g[0]))
    assert rel(gp_h, gp[0]) < 5e-4


@pytest.mark.parametrize("mode", ["density", "wavg"])
def test_p2g_cell_ordered_particles_accumulate_in_lds_and_agree_with_the_scattered_form(ops, mode):
    """particles in grid-cell order (as Styler.run sorts them) take the LDS-privatised path of the splat (a block's
    cells span a short interval of the linear index); the same particles shuffled span the whole grid and fall back to
    per-cell global atomics: both must give the oracle's grid"""
    rng = np.random.RandomState(31)
    N, G = 30000, 48
    p = np.clip(0.5 + rng.randn(N, 3) * 0.12, 0.02, 0.98).astype(np.float32)
    cell = np.floor(p * G).astype(np.int64)
    order = np.argsort((cell[:, 0] * G + cell[:, 1]) * G + cell[:, 2], kind="stable")
    x = rng.uniform(0, 1, (N, 2)).astype(np.float32)
    res, dom = [G, G, G], [float(G)] * 3
    outs = []
    for perm in (order, rng.permutation(N)):
        pp, xx = torch.tensor(p[perm]), torch.tensor(x[perm])
        if mode == "density":
            cfg = ops.make_splat_cfg(3, res, dom, 0.5, 4, 1000., 1, False, 0)
            outs.append(ops.p2g_fwd(dev(pp), cfg))
        else:
            cfg = ops.make_splat_cfg(3, res, dom, 0.5, 4, 1000., 1, False, 2)
            xs, ws = ops.p2g_fwd(dev(pp), cfg, attr=dev(xx))
            outs.append(ops.p2g_wavg_finish(xs, ws))
    pt = torch.tensor(p)[None]
    ref = O.p2g(pt, dom, res, 0.5, 1000., 1, is_2d=False, clip=False) if mode == "density" else \
        O.p2g_wavg(pt, torch.tensor(x)[None], dom, res, 0.5, 1, is_2d=False, clip=False, support=4)
    assert rel(outs[0], ref[0]) < TOL and rel(outs[1], ref[0]) < TOL
    assert rel(outs[0], outs[1]) < 1e-6
    # the gather adjoint: the block's box of the grid gradient staged in LDS (ordered) or gathered from global memory
    g = torch.tensor(rng.randn(G, G, G, 1).astype(np.float32))
    grads = []
    for perm in (order, rng.permutation(N)):
        pp, xx = torch.tensor(p[perm]), torch.tensor(x[perm])
        if mode == "density":
            gp, _, _ = ops.p2g_bwd(dev(pp), ops.make_splat_cfg(3, res, dom, 0.5, 4, 1000., 1, False, 0), dev(g))
            ga = None
        else:
            cfg = ops.make_splat_cfg(3, res, dom, 0.5, 4, 1000., 1, False, 2)
            xs, ws = ops.p2g_fwd(dev(pp), cfg, attr=dev(xx))
            g_xs, g_ws = ops.p2g_wavg_finish_bwd(xs, ws, dev(g.expand(G, G, G, 2).contiguous()))
            gp, ga, _ = ops.p2g_bwd(dev(pp), cfg, g_xs, attr=dev(xx), g_wsum=g_ws, need_p=True, need_attr=True)
            # the same adjoint in ONE launch (finish adjoint formed while the box is staged): the same lines per cell
            gp1, ga1 = ops.p2g_wavg_bwd(dev(pp), cfg, xs, ws, dev(g.expand(G, G, G, 2).contiguous()), dev(xx))
            assert torch.equal(gp1, gp) and torch.equal(ga1, ga)
        back = np.empty(N, np.int64)
        back[perm] = np.arange(N)
        grads.append((gp.cpu()[back], None if ga is None else ga.cpu()[back]))
    assert rel(grads[0][0], grads[1][0]) < 1e-4
    if mode == "wavg":
        assert rel(grads[0][1], grads[1][1]) < 1e-4
    if mode == "density":
        pt_ = torch.tensor(p)[None].requires_grad_()
        (gref,) = torch.autograd.grad(O.p2g(pt_, dom, res, 0.5, 1000., 1, is_2d=False, clip=False), pt_, g[None])
        assert rel(grads[0][0], gref[0]) < 5e-4


def test_p2g_colour_2d(ops):
    rng = np.random.RandomState(13)
    N = 400
    p = torch.tensor(rng.uniform(0, 1, (1, N, 2)), dtype=torch.float32)
    c = torch.tensor(rng.uniform(0, 1, (1, N, 3)), dtype=torch.float32).requires_grad_()
    r = torch.tensor(rng.uniform(800, 1200, (1, N, 1)), dtype=torch.float32)
    res, dom = [16, 32], [1.6, 3.2]
    ref = O.p2g(p, dom, res, 0.025, 1000., 2, pc=c, pd=r, is_2d=True, clip=False)
    cfg = ops.make_splat_cfg(2, res, dom, 0.025, 4, 1000., 2, False, 1)
    out = ops.p2g_fwd(dev(p[0]), cfg, attr=dev(c[0]), pd=dev(r[0, :, 0]))
    assert rel(out, ref[0]) < TOL
    g = torch.randn_like(ref)
    (gc,) = torch.autograd.grad(ref, c, g)
    _, gc_h, _ = ops.p2g_bwd(dev(p[0]), cfg, dev(g[0]), attr=dev(c[0]), pd=dev(r[0, :, 0]), need_p=False,
                             need_attr=True)
    assert rel(gc_h, gc[0]) < TOL


def test_p2g_wavg_3d(ops):
    rng = np.random.RandomState(14)
    N = 600
    p = torch.tensor(rng.uniform(-0.02, 1.02, (1, N, 3)), dtype=torch.float32).requires_grad_()
    x = torch.tensor(rng.uniform(0, 1, (1, N, 1)), dtype=torch.float32).requires_grad_()
    res, dom = [10, 12, 8], [10., 12., 8.]
    ref = O.p2g_wavg(p, x, dom, res, 0.5, 1, is_2d=False, clip=False, support=4)
    cfg = ops.make_splat_cfg(3, res, dom, 0.5, 4, 1000., 1, False, 2)
    xs, ws = ops.p2g_fwd(dev(p[0]), cfg, attr=dev(x[0]))
    out = ops.p2g_wavg_finish(xs, ws)
    assert rel(out, ref[0]) < TOL
    g = torch.randn_like(ref)
    gp, gx = torch.autograd.grad(ref, (p, x), g)
    g_xs, g_ws = ops.p2g_wavg_finish_bwd(xs, ws, dev(g[0]))
    gp_h, gx_h, _ = ops.p2g_bwd(dev(p[0]), cfg, g_xs, attr=dev(x[0]), g_wsum=g_ws, need_p=True, need_attr=True)
    assert rel(gx_h, gx[0]) < TOL
    assert rel(gp_h, gp[0]) < 5e-4


@pytest.mark.parametrize("shape", [(20, 9, 21), (33, 18, 37), (16, 4, 16)])
@pytest.mark.parametrize("big", [False, True])
def test_rotate_render_wave_tiles(ops, shape, big):
    """W >= 16, H >= 4: the forward march runs on 16 x 4 pixel wave tiles with the texel carry-over between
    samples; ragged tile edges, large rotations (no plane-advance reuse) and the kept rotated volume."""
    D, H, W = shape
    torch.manual_seed(9)
    d = torch.rand(1, D, H, W, 1)
    R = rots(5, 11, big=big)
    tau = 0.2
    dr = O.rotate(d, R)
    ref = O.render_unnormalised(dr, tau)
    d_rot = torch.empty(5, D, H, W, device="cuda")
    img, rs = ops.rotate_render_fwd(dev(d[0, ..., 0]), dev(R), tau, False, d_rot=d_rot)
    assert rel(d_rot, dr[..., 0]) < TOL
    assert rel(rs, dr[..., 0].sum(1)) < TOL
    assert rel(img, ref[..., 0]) < TOL
    img2, rs2 = ops.rotate_render_fwd(dev(d[0, ..., 0]), dev(R), tau, False)       # without the kept volume
    assert torch.equal(img2, img) and torch.equal(rs2, rs)
    out, _ = ops.maxnorm_fwd(img, 5)
    ref_n = torch.cat([O.render(dr[v:v + 1], tau, False) for v in range(5)])
    assert rel(out, ref_n[..., 0]) < TOL


def test_error_channel(ops):
    from neural_flow_style_amd import _lib
    with pytest.raises(RuntimeError, match="null pointer"):
        _lib.call("nfs_render_fwd", None, None, None, 1, 1, 1, 1, 0.1, 0, None)
    with pytest.raises(ValueError):
        ops.render_fwd(torch.zeros(1, 2, 2, 2), 0.1)  # CPU tensor: no fallback


@pytest.mark.parametrize("nd,C", [(2, 1), (2, 3), (3, 1), (3, 3), (3, 5)])
@pytest.mark.parametrize("linear", [False, True])
def test_g2p(ops, nd, C, linear):
    """grid -> particle sampling (cubic Catmull-Rom / linear, cell-centred, clipped cells);
    particles inside, on the border cells and outside [0,1]"""
    torch.manual_seed(31)
    dims = (9, 7) if nd == 2 else (9, 7, 11)
    g = torch.randn(1, *dims, C)
    p = torch.rand(1, 400, nd) * 1.3 - 0.15
    ref = O.g2p(g, p, is_2d=(nd == 2), is_linear=linear)[0]
    out = ops.g2p_fwd(dev(g[0]), dev(p[0]), cubic=not linear)
    assert rel(out, ref) < TOL


@pytest.mark.parametrize("mode", ["channel", "last_channel", "all", "target"])
def test_content_loss(ops, mode):
    """nfs_content_loss vs autograd of the oracle's restatement of styler_base.py:135-150 on a post-ReLU feature"""
    torch.manual_seed(17)
    B, h, w, C = 3, 5, 7, 64
    pre = torch.randn(B, h, w, C).requires_grad_()
    f = torch.relu(pre)
    ch = {"channel": 11, "last_channel": C - 1, "all": 0, "target": 5}[mode]
    tgt = torch.rand(2, h, w, C) if mode == "target" else None
    t_full = tgt[torch.arange(B) % 2] if tgt is not None else None
    wgt, amp = 2.5, 1.7
    ref = wgt * O.content_loss(f, ch, t_full, amp)
    (g_ref,) = torch.autograd.grad(ref, pre)
    loss = torch.zeros(B, device="cuda")
    g0 = torch.randn(B, h, w, C, device="cuda") * 1e-4  # accumulated into, not overwritten (same scale: no cancellation)
    g = g0.clone()
    ops.content_loss(dev(f), wgt, loss, g, channel=ch, target=dev(tgt) if tgt is not None else None, amp=amp)
    assert abs(float(loss.sum()) - float(ref)) < 1e-5 * max(abs(float(ref)), 1.0)
    assert rel(g - g0, g_ref) < 2e-5


_RR_VARIANT_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import neural_flow_style_amd.ops as ops
import neural_flow_style_amd.transform as T
rng = np.random.RandomState(3)
D, H, W = 40, 22, 37
d = torch.tensor(rng.rand(D, H, W).astype(np.float32), device="cuda")
mats = [T.rot_y_3d(t) @ T.rot_z_3d(p) for t, p in ((0, 0), (9, -4), (-35, 20), (80, 5))]
rot = T.rot_to_device(mats, "cuda")
d_rot = torch.empty(4, D, H, W, device="cuda")
img, rs = ops.rotate_render_fwd(d, rot, 0.07, False, d_rot=d_rot)
np.save(sys.argv[1], np.concatenate([img.cpu().numpy().ravel(), rs.cpu().numpy().ravel(), d_rot.cpu().numpy().ravel()]))
"""


def test_rotate_render_march_variants_agree(ops, tmp_path):
    """the ablation switches of the forward march (read once per process, hence subprocesses): plain reload of all
    four texel pairs (also the path of volumes beyond the 32-bit buffer offsets), 64 x 1 / 32 x 2 / 8 x 8 wave
    footprints -- all within f32 rounding of the default (16 x 4 tiles + carry-over)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for tag, env in (("default", {}), ("noreuse", {"NFS_RR_NOREUSE": "1"}), ("strip", {"NFS_RR_TILE": "0"}),
                     ("noreuse_strip", {"NFS_RR_NOREUSE": "1", "NFS_RR_TILE": "0"}), ("t32x2", {"NFS_RR_TILE": "5"}),
                     ("t8x8", {"NFS_RR_TILE": "3"}), ("noseg", {"NFS_RR_NOSEG": "1"}), ("view_per_xcd", {"NFS_RR_BAND": "0"})):
        f = str(tmp_path / (tag + ".npy"))
        e = dict(os.environ); e.update(env)
        subprocess.run([sys.executable, "-c", _RR_VARIANT_SCRIPT % root, f], check=True, env=e, timeout=300)
        outs[tag] = np.load(f)
    ref = outs["default"]
    for tag, o in outs.items():
        err = np.linalg.norm(o.astype(np.float64) - ref) / np.linalg.norm(ref)
        assert err < 2e-6, (tag, err)


@pytest.mark.parametrize("shape", [(2, 18, 21, 64, 128), (1, 8, 8, 128, 64), (3, 13, 30, 64, 64), (3, 20, 28, 64, 64),
                                   (2, 12, 16, 128, 128), (4, 64, 64, 128, 128),
                                   (2, 25, 25, 128, 256), (1, 50, 45, 256, 128)])    # F(5x5) layers: their own cache layout
def test_conv_relu_bit_cache_equals_float_masks(ops, shape):
    """a layer's ReLU bit cache (recorded by its forward transforms) must reproduce, bit for bit, the data gradient
    computed with the float masks x_in > 0 / x_out > 0 -- plain and pooled forms, ragged tile edges, odd sizes"""
    B, H, W, Ci, Co = shape
    torch.manual_seed(41)
    x = torch.relu(torch.randn(B, H, W, Ci, device="cuda"))               # a post-ReLU activation (zeros included)
    w = torch.randn(3, 3, Ci, Co) * 0.05
    b = torch.randn(Co, device="cuda") * 0.1
    wf, wd = ops.conv3x3_pack(dev(w), 0), ops.conv3x3_pack(dev(w), 1)
    # plain layer
    rb = ops.conv3x3_relu_bits(B, H, W, Ci, Co, False, x.device)
    assert rb is not None
    y0 = ops.conv3x3_fwd(x, wf, b, Co, True)
    y1 = ops.conv3x3_fwd(x, wf, b, Co, True, relu_bits=rb)
    assert torch.equal(y0, y1)
    gy = torch.randn(B, H, W, Co, device="cuda")
    add = torch.randn(B, H, W, Ci, device="cuda")
    g0 = ops.conv3x3_dgrad(gy, wd, Ci, x_in=x, addend=add)
    g1 = ops.conv3x3_dgrad(gy, wd, Ci, x_in=x, addend=add, relu_bits=rb)
    assert torch.equal(g0, g1)
    # an addend that has not been through the mask yet: (dgrad + addend) * mask == dgrad * mask + addend * mask
    g3 = ops.conv3x3_dgrad(gy, wd, Ci, x_in=x, addend=add, relu_bits=rb, addend_unmasked=True)
    g3_ref = ops.conv3x3_dgrad(gy, wd, Ci, x_in=x, addend=(add * (x > 0)).contiguous(), relu_bits=rb)
    assert torch.equal(g3, g3_ref)
    g2 = ops.conv3x3_dgrad(gy, wd, Ci, x_in=None, addend=add, relu_bits=rb)    # no mask asked for: the cache is ignored
    assert torch.equal(g2, ops.conv3x3_dgrad(gy, wd, Ci, x_in=None, addend=add))
    # pooled layer
    rbp = ops.conv3x3_relu_bits(B, H, W, Ci, Co, True, x.device)
    yp0, p0 = ops.conv3x3_fwd_pool(x, wf, b, Co, True)
    yp1, p1 = ops.conv3x3_fwd_pool(x, wf, b, Co, True, relu_bits=rbp)
    assert torch.equal(yp0, yp1) and torch.equal(p0, p1)
    gp = torch.randn(B, H // 2, W // 2, Co, device="cuda")
    q0 = ops.conv3x3_dgrad_pool(gp, yp0, wd, Ci, x_in=x, addend=add)
    q1 = ops.conv3x3_dgrad_pool(gp, yp1, wd, Ci, x_in=x, addend=add, relu_bits=rbp)
    assert torch.equal(q0, q1)
    q3 = ops.conv3x3_dgrad_pool(gp, yp1, wd, Ci, x_in=x, addend=add, relu_bits=rbp, addend_unmasked=True)
    q3_ref = ops.conv3x3_dgrad_pool(gp, yp1, wd, Ci, x_in=x, addend=(add * (x > 0)).contiguous(), relu_bits=rbp)
    assert torch.equal(q3, q3_ref)


def test_limb_planes_of_the_packed_filters_are_the_round_to_nearest_split_of_the_float_pack(ops):
    """nfs_conv3x3_pack writes the 16 x 16 fragment order of the transformed filters twice: as float32 (read and split in
    registers by split-limb GEMM launches of fewer than 128 rows) and as three bf16 limb planes (read ready-made by the
    larger launches).  The planes must be exactly hi = rne_bf16(x), mid = rne_bf16(x - hi), lo = rne_bf16(x - hi - mid) of the
    float pack -- then both forms feed the MFMAs the same numbers and the row threshold cannot change a result.
    Layout for Ci = Co = 64 (no F(5x5) pack): direct 9 | U 36 | Uq 36 | Uq16 36 | fused 36 | limb planes 54 floats per (ci, co)"""
    Ci = Co = 64
    n = Ci * Co
    torch.manual_seed(5)
    w = torch.randn(3, 3, Ci, Co) * torch.logspace(-6, 3, Co)[None, None, None, :]          # ten orders of magnitude
    pk = ops.conv3x3_pack(dev(w), 0).cpu().numpy()
    assert pk.size == (9 + 36 + 36 + 36 + 36 + 54) * n
    uq16 = pk[(9 + 72) * n:(9 + 108) * n]
    planes = pk[(9 + 144) * n:].view(np.uint16)
    G = 36 * (Co // 16) * (Ci // 32)
    vals = uq16.reshape(G, 2, 64, 4).transpose(0, 2, 1, 3).reshape(G, 64, 8)               # a lane's eight k of a chunk
    got = planes.reshape(G, 3, 64, 8)

    def rne_bf16(x):
        u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
        r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
        return r, (r.astype(np.uint32) << 16).view(np.float32)

    h16, h = rne_bf16(vals)
    m16, m = rne_bf16(vals - h)
    l16, _ = rne_bf16(vals - h - m)
    assert np.array_equal(got[:, 0], h16) and np.array_equal(got[:, 1], m16) and np.array_equal(got[:, 2], l16)


def test_split_limb_gemm_from_limb_planes_and_from_the_float_pack_is_bit_identical(tmp_path):
    """the same deep layers (F(5x5) and F(4x4), forward and data gradient, 8 views) in two processes: NFS_RB16S_PRE_ROWS=0
    (every split-limb launch reads the filters' limb planes) and a threshold no launch reaches (every launch splits the
    float pack in registers) -- outputs bit for bit the same"""
    import subprocess, sys, os
    code = (
        "import sys, numpy as np, torch\n"
        "import neural_flow_style_amd.ops as ops\n"
        "torch.manual_seed(3)\n"
        "outs = {}\n"
        "for name, (B, H, W, Ci, Co) in {'f5': (8, 25, 25, 256, 512), 'f4': (8, 24, 24, 256, 256)}.items():\n"
        "    x = torch.relu(torch.randn(B, H, W, Ci, device='cuda'))\n"
        "    w = torch.randn(3, 3, Ci, Co, device='cuda') * 0.03\n"
        "    b = torch.randn(Co, device='cuda') * 0.1\n"
        "    wf, wd = ops.conv3x3_pack(w, 0), ops.conv3x3_pack(w, 1)\n"
        "    y = ops.conv3x3_fwd(x, wf, b, Co, True)\n"
        "    g = ops.conv3x3_dgrad(torch.randn(B, H, W, Co, device='cuda'), wd, Ci, x_in=x)\n"
        "    outs[name + '_y'] = y.cpu().numpy(); outs[name + '_g'] = g.cpu().numpy()\n"
        "np.savez(sys.argv[1], **outs)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for tag, rows in (("planes", "0"), ("registers", "1000000000")):
        out = str(tmp_path / (tag + ".npz"))
        env = dict(os.environ, NFS_RB16S_PRE_ROWS=rows, NFS_GEMM_TUNE="0", PYTHONPATH=root)
        subprocess.run([sys.executable, "-c", code, out], check=True, env=env, cwd=root, timeout=600)
        res.append(np.load(out))
    for k in res[0].files:
        assert np.array_equal(res[0][k], res[1][k]), k


@pytest.mark.parametrize("H,W,Ci,Co", [(25, 25, 256, 512), (24, 29, 512, 256), (50, 50, 256, 256),
                                       (24, 24, 512, 512), (22, 27, 256, 256)])             # the last two: F(4x4)
def test_transforms_of_few_and_of_many_tiles_agree(ops, H, W, Ci, Co):
    """the F(5x5) / F(4x4) transforms run as seven / six waves per (tile, 64 / 128 channels) up to 65536 (tile, channel)
    items and as one thread per (tile, channel / channel pair) beyond (NFS_W5_WAVES7_MAX, NFS_W4_WAVES6_MAX): the same
    image alone and stacked 12 times -- outputs and data gradients per image agree to rounding (the GEMM between them may
    split K differently), the ReLU bit caches exactly"""
    torch.manual_seed(H + Ci)
    reps = 12 if H < 50 else 4
    assert H // 5 * (W // 5) * max(Ci, Co) <= 65536 < reps * ((H + 4) // 5) * ((W + 4) // 5) * min(Ci, Co)
    x1 = torch.relu(torch.randn(1, H, W, Ci, device="cuda"))
    w = torch.randn(3, 3, Ci, Co) * (2.0 / (9 * Ci)) ** 0.5
    b = torch.randn(Co, device="cuda") * 0.1
    wf, wd = ops.conv3x3_pack(dev(w), 0), ops.conv3x3_pack(dev(w), 1)
    gy1 = torch.randn(1, H, W, Co, device="cuda")
    add1 = torch.randn(1, H, W, Ci, device="cuda")
    res = []
    for B in (1, reps):
        x, gy, add = (t.repeat(B, 1, 1, 1).contiguous() for t in (x1, gy1, add1))
        rb = ops.conv3x3_relu_bits(B, H, W, Ci, Co, False, x.device)
        y = ops.conv3x3_fwd(x, wf, b, Co, True, relu_bits=rb)
        g = ops.conv3x3_dgrad(gy, wd, Ci, x_in=x, addend=add, relu_bits=rb)
        res.append((y, g, rb.view(B, -1).clone()))
    (y1, g1, r1), (yb, gb, rbb) = res
    for i in (0, reps - 1):
        assert torch.equal(r1[0].view(torch.int32), rbb[i].view(torch.int32))
        for a, c in ((y1[0], yb[i]), (g1[0], gb[i])):
            assert float((a - c).norm()) <= 1e-5 * float(a.norm())            # (F(5x5) against float64: ~5e-6, budget 2e-5)
            assert float((a - c).abs().max()) <= 1e-4 * float(a.abs().max())


@pytest.mark.parametrize("shape", [(2, 18, 22, 64, 128), (2, 20, 24, 64, 128)])     # three-kernel / single-kernel path
def test_pooled_layer_without_full_resolution_output(ops, shape):
    """with the bit cache a pooled layer need not write its full-resolution output: forward (pool only) and data
    gradient (x_out = None) agree bit for bit with the materialised form"""
    B, H, W, Ci, Co = shape
    torch.manual_seed(43)
    x = torch.relu(torch.randn(B, H, W, Ci, device="cuda"))
    w = torch.randn(3, 3, Ci, Co) * 0.05
    b = torch.randn(Co, device="cuda") * 0.1
    wf, wd = ops.conv3x3_pack(dev(w), 0), ops.conv3x3_pack(dev(w), 1)
    rb0 = ops.conv3x3_relu_bits(B, H, W, Ci, Co, True, x.device)
    rb1 = ops.conv3x3_relu_bits(B, H, W, Ci, Co, True, x.device)
    y0, p0 = ops.conv3x3_fwd_pool(x, wf, b, Co, True, relu_bits=rb0)
    y1, p1 = ops.conv3x3_fwd_pool(x, wf, b, Co, True, relu_bits=rb1, want_y=False)
    assert y1 is None and torch.equal(p0, p1)
    assert torch.equal(rb0.view(torch.int32), rb1.view(torch.int32))
    gp = torch.randn(B, H // 2, W // 2, Co, device="cuda")
    q0 = ops.conv3x3_dgrad_pool(gp, y0, wd, Ci, x_in=x, relu_bits=rb0)
    q1 = ops.conv3x3_dgrad_pool(gp, None, wd, Ci, x_in=x, relu_bits=rb1, hw=(H, W))
    assert torch.equal(q0, q1)
    with pytest.raises(AssertionError):                           # neither the output nor the cache: refused by the binding
        ops.conv3x3_dgrad_pool(gp, None, wd, Ci, x_in=x, relu_bits=None, hw=(H, W))


@pytest.mark.parametrize("B,H,W,Ci,Co,floor", [
    (2, 24, 20, 64, 128, 5e-4),        # 64-row tiles of the round-2 kernel (K = 64: no 16-row instance applies? it does: rb16s)
    (8, 25, 25, 512, 512, 5e-3),       # conv4_2 at 8 views: F(5x5), 200 rows (the 208-row / 112-row split-limb tiles)
    (2, 25, 25, 256, 512, 5e-3),       # 50 rows, K != N
    (1, 50, 50, 256, 256, 5e-3),       # conv3_2 at one view: 100 rows
    (8, 12, 12, 512, 512, 5e-4),       # conv5_1 at 8 views: F(4x4), 72 rows (80-row tile, 5 live)
    (1, 12, 12, 512, 512, 5e-4),       # 9 rows, two K parts
    (3, 28, 28, 256, 256, 5e-4)])      # F(4x4), 147 rows
def test_split_limb_gemm_is_float32_accurate(B, H, W, Ci, Co, floor):
    """The split-limb GEMM mode (nfs_gemm_mode(1): every float32 operand written exactly as three bf16 limbs, six limb
    products on the bf16 MFMA, float32 accumulation) against a float64 convolution, next to the float32-input MFMA
    (mode 0) on the same inputs: same accuracy class -- each product is carried to 2^-26, below float32's rounding unit.
    Inputs span 10 orders of magnitude per tensor (the limb split must be exact at every scale).  The shapes cover the
    16-row register-B instance (rb16s: F(5x5) and F(4x4) deep layers, ragged last row tiles, K parts)."""
    from neural_flow_style_amd import ops
    rng = np.random.RandomState(12)
    x = (rng.randn(B, H, W, Ci) * np.exp(rng.uniform(-11, 11, (B, H, W, Ci)))).astype(np.float32)
    x = np.maximum(x, 0)
    w = (rng.randn(3, 3, Ci, Co) * 0.05 * np.exp(rng.uniform(-3, 3, (3, 3, Ci, Co)))).astype(np.float32)
    b = rng.randn(Co).astype(np.float32)
    ref = torch.nn.functional.conv2d(torch.tensor(x).double().permute(0, 3, 1, 2),
                                     torch.tensor(w).double().permute(3, 2, 0, 1), torch.tensor(b).double(),
                                     padding=1).permute(0, 2, 3, 1)
    scale = torch.nn.functional.conv2d(torch.tensor(x).double().abs().permute(0, 3, 1, 2),
                                       torch.tensor(w).double().abs().permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
    xd, wd, bd = torch.tensor(x).cuda(), torch.tensor(w).cuda(), torch.tensor(b).cuda()
    pk = ops.conv3x3_pack(wd, 0)
    errs = {}
    prev = ops.gemm_mode(None)
    try:
        for mode in (0, 1):
            ops.gemm_mode(mode)
            y = ops.conv3x3_fwd(xd, pk, bd, Co, relu=False).double().cpu()
            # error relative to the magnitude of the terms summed (the natural scale of float32 rounding)
            errs[mode] = float(((y - ref).abs() / (scale + 1e-300)).max())
    finally:
        ops.gemm_mode(prev)
    # the error floor here is Winograd's, not the GEMM's: F(4x4)'s transforms mix pixels whose magnitudes differ by
    # orders of magnitude in this input (measured 1.1e-4 for the float32-input MFMA; F(5x5)'s larger constants put it
    # higher); the point is that the split-limb arithmetic lands on the SAME floor
    print("split-limb accuracy %s: f32-input MFMA %.3g, split-limb %.3g" % ((B, H, W, Ci, Co), errs[0], errs[1]))
    assert errs[0] < floor and errs[1] < floor, errs
    assert errs[1] < 1.5 * errs[0] + 1e-7, errs


@pytest.mark.parametrize("magnitude", [1e28, 1e-28])
def test_split_limb_gemm_at_the_ends_of_the_float_range(magnitude):
    """bf16 shares float32's exponent range, so the three-limb split is exact wherever all three limbs are normal
    numbers: activations of uniform magnitude 1e28 (products ~1e27, far from overflow through the Winograd transforms'
    constants) and 1e-28 (the lowest limb ~1e-33, still normal) must come out as accurately as at unit scale -- no inf - inf
    from a rounded-up leading limb, no flushed low limb.  (Beyond ~3e38 / 2^8 the leading limb can round to inf and below
    ~1e-33 the low limbs flush: both outside anything a loss network sees; stated in INTEGRATION.md.)"""
    from neural_flow_style_amd import ops
    rng = np.random.RandomState(14)
    B, H, W, Ci, Co = 2, 25, 25, 256, 256
    x = (np.abs(rng.randn(B, H, W, Ci)) * magnitude).astype(np.float32)
    w = (rng.randn(3, 3, Ci, Co) * 0.05).astype(np.float32)
    ref = torch.nn.functional.conv2d(torch.tensor(x).double().permute(0, 3, 1, 2),
                                     torch.tensor(w).double().permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
    scale = torch.nn.functional.conv2d(torch.tensor(x).double().abs().permute(0, 3, 1, 2),
                                       torch.tensor(w).double().abs().permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
    xd, wd = torch.tensor(x).cuda(), torch.tensor(w).cuda()
    pk = ops.conv3x3_pack(wd, 0)
    errs = {}
    prev = ops.gemm_mode(None)
    try:
        for mode in (0, 1):
            ops.gemm_mode(mode)
            y = ops.conv3x3_fwd(xd, pk, None, Co, relu=False)
            assert torch.isfinite(y).all()
            errs[mode] = float(((y.double().cpu() - ref).abs() / scale).max())
    finally:
        ops.gemm_mode(prev)
    assert errs[0] < 2e-5 and errs[1] < 2e-5, errs
    assert errs[1] < 1.5 * errs[0] + 1e-7, errs


def test_gradient_chain_identical_in_both_gemm_modes():
    """VGG forward + data-gradient chain with the Winograd GEMMs in split-limb mode vs float32-input MFMA mode: the
    two float32-equivalent arithmetics agree to float32 rounding (1e-5), far inside the 1e-3 bar"""
    from neural_flow_style_amd import ops, vgg
    rng = np.random.RandomState(13)
    net = vgg.VGG(vgg.synthetic_weights(123, upto="conv4_1"), "cuda")
    x = torch.tensor(rng.uniform(-120, 130, (2, 40, 40, 3)).astype(np.float32)).cuda()
    out = {}
    prev = ops.gemm_mode(None)
    try:
        for mode in (0, 1):
            ops.gemm_mode(mode)
            acts = net.forward(x, "conv4_1")
            g_top = torch.ones_like(acts["conv4_1"]) * (acts["conv4_1"] > 0)
            gx = net.backward(acts, {"conv4_1": g_top.contiguous()}, "conv4_1")
            out[mode] = (acts["conv4_1"].clone(), gx.clone())
    finally:
        ops.gemm_mode(prev)
    assert rel(out[1][0], out[0][0]) < 1e-5 and rel(out[1][1], out[0][1]) < 1e-5


def test_hist_loss_matches_the_oracle_restatement():
    """nfs_hist_loss vs oracle.hist_loss (util.histogram_match_tf restated in NumPy with SciPy's interp1d): loss and
    gradient for a post-ReLU-like feature (many exact zeros) and for an image-like 'input' layer; the matching is a
    staircase, so a value within float rounding of a bin edge may take the neighbouring step: a handful of elements"""
    from neural_flow_style_amd import ops
    rng = np.random.RandomState(21)
    for (B, h, w, C, Bt, ht, wt, relu) in ((2, 13, 11, 5, 1, 9, 14, True), (1, 20, 20, 3, 1, 20, 20, False)):
        f = rng.gamma(2.0, 15.0, (B, h, w, C)).astype(np.float32)
        t = (rng.rand(Bt, ht, wt, C) * 180 + rng.rand(1, 1, 1, C) * 40).astype(np.float32)
        if relu:
            f[rng.rand(*f.shape) < 0.3] = 0.0
        ft = torch.tensor(f, requires_grad=True)
        lo = O.hist_loss(ft, torch.tensor(t))
        (go,) = torch.autograd.grad(lo * 0.7, ft)
        if relu:
            go = go * (ft.detach() > 0)
        loss = torch.zeros(B, device="cuda")
        g = torch.zeros(B, h, w, C, device="cuda")
        ops.hist_loss(torch.tensor(f).cuda(), torch.tensor(t).cuda(), 0.7, loss, g, relu_mask=relu)
        assert abs(float(loss.sum()) - 0.7 * float(lo)) < 2e-3 * 0.7 * float(lo)
        d = (g.cpu() - go).abs()
        assert float((d > 1e-3 * go.abs().max()).float().mean()) < 5e-3
        assert rel(g.cpu(), go) < 3e-2


def test_render_max_and_mean_ray_modes():
    """nfs_render_fwd / _bwd with mode 2 (reduce_max along the ray, gradient split equally among ties like TF's) and 3
    (reduce_mean) against autograd, in place on the volume as the engine calls them"""
    from neural_flow_style_amd import ops
    rng = np.random.RandomState(8)
    V, D, H, W = 2, 9, 7, 6
    d = rng.rand(V, D, H, W).astype(np.float32)
    d[0, 2, 3, 4] = d[0, 7, 3, 4] = 1.5                      # a two-way tie at the maximum of one ray
    g = rng.randn(V, H, W).astype(np.float32)
    for mode, fn in ((2, lambda x: x.amax(dim=1)), (3, lambda x: x.mean(dim=1))):
        dt = torch.tensor(d, requires_grad=True)
        want = fn(dt)
        (gd_ref,) = torch.autograd.grad((want * torch.tensor(g)).sum(), dt)
        dv = torch.tensor(d).cuda()
        img, rs = ops.render_fwd(dv, 0.3, mode)
        assert rel(img.cpu(), want.detach()) < 1e-6
        gd, gmax = ops.render_bwd(dv, rs, torch.tensor(g).cuda(), 0.3, mode, g_d=dv, want_max=True)     # in place
        assert rel(gd.cpu(), gd_ref) < 1e-6
        assert abs(float(gmax) - float(gd_ref.abs().max())) < 1e-6
    assert float(gd_ref.sum()) != 0.0
    dv = torch.tensor(d).cuda()
    _, rs = ops.render_fwd(dv, 0.3, 2)
    gd = ops.render_bwd(dv, rs, torch.tensor(g).cuda(), 0.3, 2)
    assert float(gd[0, 2, 3, 4]) == float(gd[0, 7, 3, 4]) == 0.5 * g[0, 3, 4]


def test_maxnorm_input_fused_equals_the_two_step_form():
    """nfs_maxnorm_input_fwd / _bwd (max-normalisation + loss-net input in one pass each way) against nfs_maxnorm_fwd +
    nfs_loss_net_input_fwd and nfs_loss_net_input_bwd + nfs_maxnorm_bwd: the same arithmetic value for value, incl. a
    tie at the maximum; one group per view and one group over all views"""
    from neural_flow_style_amd import ops
    rng = np.random.RandomState(2)
    V, H, W = 4, 150, 140
    img = rng.rand(V, H, W).astype(np.float32)
    img[1, 3, 5] = img[1, 70, 9] = 2.0                        # a tie at the maximum of view 1
    g_x = rng.randn(V, H, W, 3).astype(np.float32)
    it, gt = torch.tensor(img).cuda(), torch.tensor(g_x).cuda()
    for groups in (V, 1, 2):
        norm, gmax = ops.maxnorm_fwd(it, groups)
        _, x_ref = ops.loss_net_input_fwd(norm.unsqueeze(-1), H, W, want_d_img=False)
        g_norm = ops.loss_net_input_bwd(gt, H, W, 1).reshape(V, H, W)
        gi_ref = ops.maxnorm_bwd(it, gmax, g_norm)
        x, gmax2 = ops.maxnorm_input_fwd(it, groups)
        gi = ops.maxnorm_input_bwd(it, gmax2, gt)
        torch.cuda.synchronize()
        # (the same operations; the compiler may contract a multiply-add in one form and not in the other: last-bit)
        assert torch.equal(gmax, gmax2) and float((x - x_ref).abs().max()) <= 4e-5          # |x| <= 255: 2 ulp
        assert rel(gi.cpu(), gi_ref.cpu()) < 1e-6


def test_gram_style_group_equals_the_per_layer_chain():
    """nfs_gram_style_group_fwd + nfs_gram_group_bwd (all style layers in three launches, the style loss as per-block
    partial sums) against the per-layer entry points nfs_gram_fwd -> nfs_style_loss_fwd -> nfs_gram_bwd: slab layers,
    one-slab layers, diagonal-only (C = 64) and off-diagonal tile pairs, ragged last chunks, masked and unmasked dF"""
    from neural_flow_style_amd import ops
    rng = np.random.RandomState(5)
    B = 3
    shapes = [(40, 40, 64), (37, 35, 128), (35, 36, 256), (6, 5, 512), (3, 3, 512), (24, 25, 64)]
    Fs, Gss, ws, masks = [], [], [], []
    for i, (h, w, C) in enumerate(shapes):
        F = np.maximum(rng.randn(B, h, w, C), 0).astype(np.float32)
        Fs.append(torch.tensor(F).cuda())
        S = np.maximum(rng.randn(1, h, w, C), 0).astype(np.float32)
        Gss.append(ops.gram_fwd(torch.tensor(S).cuda(), 1.0 / (2.0 * h * w * C)))
        ws.append(0.5 + 0.25 * i)
        masks.append(i % 2 == 0)
    parts, dFs, Gs_out = ops.gram_style_group(Fs, Gss, ws, masks, want_G=True)
    torch.cuda.synchronize()
    assert torch.isfinite(parts).all()
    loss_ref = torch.zeros(B, device="cuda")
    for F, Gs, w, m, dF, G in zip(Fs, Gss, ws, masks, dFs, Gs_out):
        _, h, w_, C = F.shape
        sc = 1.0 / (2.0 * h * w_ * C)
        G_ref = ops.gram_fwd(F, sc)
        assert rel(G.cpu(), G_ref.cpu()) < 1e-6
        assert float((G - G.transpose(1, 2)).abs().max()) == 0.0          # mirrored tiles are copies
        Dm = ops.style_loss_fwd(G_ref, Gs, w, loss_ref)
        dF_ref = ops.gram_bwd(F, Dm, sc, relu_mask=m)
        assert rel(dF.cpu(), dF_ref.cpu()) < 2e-5, (h, w_, C)
    got = parts.sum(0)
    assert rel(got.cpu(), loss_ref.cpu()) < 1e-5
    # deterministic: no atomics anywhere in the grouped chain
    parts2, dFs2, _ = ops.gram_style_group(Fs, Gss, ws, masks)
    assert torch.equal(parts, parts2) and all(torch.equal(a, b) for a, b in zip(dFs, dFs2))


def test_hist_loss_masked_branch_flat_channels_and_empty_masks():
    """the masked branch (styler_base.py:104-125, 196-201: tf.boolean_mask of the source by mask != 0) against the
    oracle, and the cases the reference leaves undefined -- a flat channel (max == min over source and template) and a
    source with every pixel masked out -- which this build defines as "skipped": loss 0, gradient 0, no NaN"""
    from neural_flow_style_amd import ops
    rng = np.random.RandomState(33)
    B, h, w, C = 2, 14, 10, 4
    f = rng.gamma(2.0, 15.0, (B, h, w, C)).astype(np.float32)
    f[rng.rand(*f.shape) < 0.3] = 0.0
    t = (rng.rand(1, 9, 12, C) * 180).astype(np.float32)
    m = (rng.rand(B, h, w, 1) * (rng.rand(B, h, w, 1) < 0.6)).astype(np.float32)     # 40 % exact zeros
    f[..., 2] = 7.5; t[..., 2] = 7.5                       # flat channel
    m[1] = 0.0                                             # image 1: nothing left to match
    ft = torch.tensor(f, requires_grad=True)
    lo = O.hist_loss(ft, torch.tensor(t), mask=torch.tensor(m))
    (go,) = torch.autograd.grad(lo * 0.7, ft)
    go = go * (ft.detach() > 0)
    loss = torch.zeros(B, device="cuda")
    g = torch.zeros(B, h, w, C, device="cuda")
    ops.hist_loss(torch.tensor(f).cuda(), torch.tensor(t).cuda(), 0.7, loss, g, relu_mask=True,
                  mask=torch.tensor(m).cuda().contiguous())
    assert torch.isfinite(loss).all() and torch.isfinite(g).all()
    assert float(loss[1]) == 0.0 and float(g[1].abs().max()) == 0.0          # empty mask
    assert float(g[..., 2].abs().max()) == 0.0                                # flat channel
    assert float(g[0][torch.tensor(m[0, ..., 0]) == 0].abs().max()) == 0.0    # masked-out pixels carry no gradient
    assert abs(float(loss.sum()) - 0.7 * float(lo)) < 2e-3 * 0.7 * float(lo)
    d = (g.cpu() - go).abs()
    assert float((d > 1e-3 * go.abs().max()).float().mean()) < 5e-3
    assert rel(g.cpu(), go) < 3e-2
    # unmasked entry point, flat channel: same definition
    loss2 = torch.zeros(B, device="cuda")
    g2 = torch.zeros(B, h, w, C, device="cuda")
    ops.hist_loss(torch.tensor(f).cuda(), torch.tensor(t).cuda(), 1.0, loss2, g2, relu_mask=False)
    assert torch.isfinite(loss2).all() and float(g2[..., 2].abs().max()) == 0.0
    lo2 = O.hist_loss(torch.tensor(f), torch.tensor(t))
    assert abs(float(loss2.sum()) - float(lo2)) < 2e-3 * float(lo2)


@pytest.mark.parametrize("with_mask", [False, True])
def test_hist_loss_pixel_parallel_form_equals_the_per_channel_kernel(with_mask):
    """images of <= 4 channels (the default hist layer, the 3-channel loss-net input) take nfs_hist_loss_wide -- pixels
    spread over the chip, per-channel state in a workspace -- where one block per (image, channel) would run 3 blocks:
    the same bins, table and matched values, so the SAME gradient bit for bit and the same loss to summation order;
    two images of 150 x 225, one template of another size, a flat channel in image 1"""
    from neural_flow_style_amd import _lib, ops
    gen = torch.Generator(device="cuda").manual_seed(5)
    B, h, w, C = 2, 150, 225, 3
    f = (torch.rand(B, h, w, C, device="cuda", generator=gen) * 255).contiguous()
    f[1, ..., 1] = 12.0
    t = (torch.rand(1, 120, 200, C, device="cuda", generator=gen) * 200 + 20).contiguous()
    m = None
    if with_mask:
        m = (torch.rand(B, h, w, device="cuda", generator=gen) > 0.35).float().contiguous()
    l_w, g_w = torch.zeros(B, device="cuda"), torch.zeros_like(f)
    ops.hist_loss(f, t, 0.3, l_w, g_w, relu_mask=False, mask=m)
    l_c, g_c = torch.zeros(B, device="cuda"), torch.zeros_like(f)
    _lib.call("nfs_hist_loss_masked", ops._ptr(f), ops._ptr(t), ops._ptr(m), ops._ptr(l_c), ops._ptr(g_c), B, 1, h * w,
              120 * 200, C, 0.3, 0, ops._stream())
    assert torch.equal(g_w, g_c) and float(g_w.abs().max()) > 0
    assert torch.allclose(l_w, l_c, rtol=1e-5)
    l_2, g_2 = torch.zeros(B, device="cuda"), torch.zeros_like(f)       # deterministic: a second call, the same bits
    ops.hist_loss(f, t, 0.3, l_2, g_2, relu_mask=False, mask=m)
    assert torch.equal(l_2, l_w) and torch.equal(g_2, g_w)


def test_style_mask_kernels():
    """legacy bicubic resize of the density mask, masked features with the 2*area*C denominator, masked gradient
    (styler_base.py:165-173) against the oracle's restatement"""
    from neural_flow_style_amd import ops
    rng = np.random.RandomState(41)
    x = rng.rand(2, 17, 12, 1).astype(np.float32)
    for oh, ow in ((8, 6), (17, 12), (25, 30), (4, 3)):
        want = O.tf1_resize_bicubic(torch.tensor(x), oh, ow)
        got = ops.resize_bicubic_tf1(torch.tensor(x).cuda(), oh, ow).cpu()
        assert float((got - want).abs().max()) < 2e-6, (oh, ow)
    F_ = np.maximum(rng.randn(2, 8, 6, 16), 0).astype(np.float32)
    m = O.tf1_resize_bicubic(torch.tensor(x), 8, 6)
    Fm, scale = ops.style_mask_apply(torch.tensor(F_).cuda(), m.cuda().contiguous())
    assert rel(Fm.cpu(), torch.tensor(F_) * m) < 1e-6
    assert rel(scale.cpu(), 1.0 / (2.0 * m[..., 0].sum(dim=(1, 2)) * 16)) < 1e-6
    g = rng.randn(2, 8, 6, 16).astype(np.float32)
    got = ops.style_mask_bwd(torch.tensor(g).cuda(), m.cuda().contiguous(), torch.tensor(F_).cuda()).cpu()
    assert rel(got, torch.tensor(g) * m * (torch.tensor(F_) > 0)) < 1e-6


@pytest.mark.parametrize("D,H,W,world", [(11, 3, 4, 3), (27, 6, 6, 8), (24, 8, 8, 8), (40, 10, 10, 4), (5, 4, 4, 8),
                                         (200, 20, 20, 8)])
def test_slab_pack_is_the_index_table_gather(D, H, W, world):
    """send buffer of the D-slab reduce-scatter (nfs_slab_pack): overlapping plane ranges of the padded gradient volume,
    zero past it (ragged / empty slabs), the loss plane behind every chunk -- bit for bit the gather through
    parallel.slab_pack_index that it replaced"""
    from neural_flow_style_amd import ops, parallel
    cs, _ = parallel.slab_plan(D, world)
    rng = np.random.RandomState(D + world)
    gpad = torch.zeros(D + 5, H, W, device="cuda")
    gpad[2:D + 2] = torch.tensor(rng.randn(D, H, W).astype(np.float32)).cuda()
    gpad[D + 4].view(-1)[0] = 7.25
    pack = torch.full((world, cs + 5, H, W), float("nan"), device="cuda")
    ops.slab_pack(gpad, pack, D, world, cs)
    idx = torch.tensor(parallel.slab_pack_index(D, world), device="cuda")
    assert torch.equal(pack.view(-1, H, W), gpad.index_select(0, idx))
