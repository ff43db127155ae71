"""the Inception-v1 loss network (``tensorflow_inception_graph.pb`` of styler_base.py:17-30, 51-57) --
node kernels (csrc/inception.hip) against the oracle's TF-semantics restatements, the assembled network against
``oracle.inception_v1_features`` + autograd, the engine gradient with the reference driver's style layers
('conv2d2', 'mixed3b', 'mixed4b': test_smokegun.py:141) and run.bat's '*_pre_relu' content layers, and the
``Styler(config).run`` surface with ``network='tensorflow_inception_graph.pb'``."""
import numpy as np
import pytest
import torch

from oracle import nfs_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def nchw(x):
    return torch.as_tensor(x).permute(0, 3, 1, 2)


def nhwc(x):
    return x.permute(0, 2, 3, 1)


# (k, stride, Cin, Cout, B, H, W): 1x1 / 3x3 / 5x5 of the modules incl. widths that are no multiple of 16, the 7x7
# first layer at both strides, odd and even sizes (SAME pads differ), more than one 128-row tile, a single pixel
CONV_CASES = [(1, 1, 192, 64, 2, 9, 11), (3, 1, 96, 128, 2, 10, 7), (5, 1, 16, 32, 1, 12, 12), (3, 1, 24, 204, 2, 5, 6),
              (1, 1, 508, 112, 1, 4, 5), (5, 1, 48, 128, 3, 7, 9), (7, 2, 3, 64, 2, 21, 30), (7, 1, 3, 64, 1, 13, 8),
              (7, 2, 3, 64, 1, 64, 96), (3, 1, 64, 192, 4, 40, 56), (1, 1, 832, 384, 1, 1, 1), (3, 2, 32, 64, 1, 9, 10)]


@pytest.mark.parametrize("k,stride,ci,co,B,H,W", CONV_CASES)
def test_conv2d_same_matches_tf_semantics(k, stride, ci, co, B, H, W):
    from neural_flow_style_amd import ops
    rng = np.random.RandomState(k * 100 + ci)
    x = rng.randn(B, H, W, ci).astype(np.float32)
    w = (rng.randn(k, k, ci, co) / np.sqrt(k * k * ci)).astype(np.float32)
    b = rng.randn(co).astype(np.float32)
    pre = nhwc(O.tf_conv2d_same(nchw(x).double(), torch.tensor(w).double(), torch.tensor(b).double(), stride))
    Ho, Wo = pre.shape[1], pre.shape[2]
    # operands are channel ranges of wider rows; the rest of the rows must stay untouched
    ldx = ci if ci <= 4 else ci + 8
    cx = 0 if ci <= 4 else 4
    xb = torch.full((B, H, W, ldx), 7.0, device=DEV)
    xb[..., cx:cx + ci] = torch.tensor(x)
    yb = torch.full((B, Ho, Wo, co + 12), -3.0, device=DEV)
    pb = torch.full((B, Ho, Wo, co + 4), 5.0, device=DEV)
    packed = ops.conv2d_pack(torch.tensor(w, device=DEV))
    ops.conv2d_fwd(xb, cx, ci, packed, torch.tensor(b, device=DEV), yb, 8, co, k, k, stride, relu=True, y_pre=pb, cp=4)
    assert rel(pb[..., 4:], pre) < 2e-6
    assert rel(yb[..., 8:8 + co], pre.clamp_min(0)) < 2e-6
    assert float((yb[..., :8] + 3).abs().max()) == 0 and float((yb[..., 8 + co:] + 3).abs().max()) == 0
    assert float((pb[..., :4] - 5).abs().max()) == 0
    # accumulate, no bias, no ReLU
    before = yb.clone()
    ops.conv2d_fwd(xb, cx, ci, packed, None, yb, 8, co, k, k, stride, relu=False, accumulate=True)
    want = before[..., 8:8 + co].double().cpu() + (pre - torch.tensor(b).double())
    assert rel(yb[..., 8:8 + co], want) < 2e-6


@pytest.mark.parametrize("k,ci,co,B,H,W", [(1, 192, 64, 2, 9, 11), (3, 96, 128, 2, 10, 7), (5, 16, 32, 1, 12, 12),
                                           (3, 24, 204, 2, 5, 6), (1, 508, 112, 1, 4, 5), (5, 48, 128, 3, 7, 9)])
def test_conv2d_data_gradient_is_the_same_kernel_on_transposed_filters(k, ci, co, B, H, W):
    """gx = dgrad(gy * (act > 0)): masked while loaded, accumulated into the input gradient"""
    from neural_flow_style_amd import ops
    rng = np.random.RandomState(k + ci)
    w = (rng.randn(k, k, ci, co) / np.sqrt(k * k * ci)).astype(np.float32)
    gy = rng.randn(B, H, W, co).astype(np.float32)
    act = rng.randn(B, H, W, co).astype(np.float32)
    x = torch.zeros(B, ci, H, W, dtype=torch.float64, requires_grad=True)
    y = O.tf_conv2d_same(x, torch.tensor(w).double())
    (y * nchw(gy * (act > 0)).double()).sum().backward()
    want = nhwc(x.grad)
    gx = torch.full((B, H, W, ci + 4), 2.0, device=DEV)
    dg = ops.conv2d_pack(torch.tensor(w, device=DEV), transpose=True)
    ops.conv2d_fwd(torch.tensor(gy, device=DEV), 0, co, dg, None, gx, 4, ci, k, k, 1, relu=False,
                   x_mask=torch.tensor(act, device=DEV), accumulate=True)
    assert rel(gx[..., 4:] - 2.0, want) < 3e-6
    assert float((gx[..., :4] - 2).abs().max()) == 0


@pytest.mark.parametrize("stride,B,H,W", [(2, 2, 21, 30), (2, 1, 16, 16), (1, 1, 13, 8), (2, 1, 64, 97)])
def test_conv2d_small_data_gradient_down_to_the_image(stride, B, H, W):
    from neural_flow_style_amd import ops
    rng = np.random.RandomState(stride + H)
    w = (rng.randn(7, 7, 3, 64) / np.sqrt(147)).astype(np.float32)
    x = torch.zeros(B, 3, H, W, dtype=torch.float64, requires_grad=True)
    y = O.tf_conv2d_same(x, torch.tensor(w).double(), stride=stride)
    gy = rng.randn(*nhwc(y).shape).astype(np.float32)
    act = rng.randn(*nhwc(y).shape).astype(np.float32)
    (y * nchw(gy * (act > 0)).double()).sum().backward()
    got = ops.conv2d_dgrad_small(torch.tensor(gy, device=DEV), 0, 64, torch.tensor(w, device=DEV), (H, W), stride,
                                 y_act=torch.tensor(act, device=DEV))
    assert rel(got, nhwc(x.grad)) < 3e-6


@pytest.mark.parametrize("stride,B,H,W,C", [(2, 2, 9, 12, 64), (2, 1, 10, 11, 192), (1, 2, 7, 5, 256), (1, 1, 1, 1, 64),
                                            (2, 1, 2, 3, 8)])
def test_maxpool3_same_and_its_adjoint(stride, B, H, W, C):
    from neural_flow_style_amd import ops
    rng = np.random.RandomState(stride * 10 + H)
    x = rng.randn(B, H, W, C).astype(np.float32)
    xt = nchw(x).double().requires_grad_()
    y = O.tf_maxpool3_same(xt, stride)
    gy = rng.randn(*nhwc(y).shape).astype(np.float32)
    (y * nchw(gy).double()).sum().backward()
    got, arg = ops.maxpool3_fwd(torch.tensor(x, device=DEV), stride)
    assert torch.equal(got.cpu(), nhwc(y).float())
    base = torch.full((B, H, W, C), 0.5, device=DEV)
    gx = ops.maxpool3_bwd(torch.tensor(gy, device=DEV), arg, (H, W), stride, gx=base)
    assert rel(gx - 0.5, nhwc(xt.grad)) < 1e-6
    gx2 = ops.maxpool3_bwd(torch.tensor(gy, device=DEV), arg, (H, W), stride)
    assert rel(gx2, nhwc(xt.grad)) < 1e-6
    # ... handing the gradient on with the pooled tensor's own ReLU adjoint
    gx3 = ops.maxpool3_bwd(torch.tensor(gy, device=DEV), arg, (H, W), stride, gx=base.clone().fill_(0.5),
                           relu_of=torch.tensor(x, device=DEV))
    assert rel(gx3, (nhwc(xt.grad) + 0.5) * torch.tensor(x > 0)) < 1e-6


def test_maxpool3_gives_the_gradient_to_the_first_maximum():
    """ties: the first maximum in row-major window order takes the gradient (TF's CPU kernel; PyTorch's too)"""
    from neural_flow_style_amd import ops
    x = torch.zeros(1, 3, 3, 4, device=DEV)
    y, arg = ops.maxpool3_fwd(x, 1)
    g = ops.maxpool3_bwd(torch.ones_like(y), arg, (3, 3), 1)
    # window of output (0,0) = rows 0..1 x cols 0..1 -> first in-range tap is input (0,0); ... every window's first
    # in-range tap: outputs in row 0 / col 0 start at the border
    assert float(g.sum()) == 9 * 4 and float(g[0, 0, 0, 0]) == 4.0


@pytest.mark.parametrize("C,ld,r,bias,alpha,beta", [(64, 64, 5, 2.0, 1e-4, 0.5), (192, 192, 5, 2.0, 1e-4, 0.5),
                                                   (24, 64, 2, 1.0, 2e-2, 0.75)])
def test_lrn_and_its_adjoint(C, ld, r, bias, alpha, beta):
    from neural_flow_style_amd import ops
    rng = np.random.RandomState(C)
    x = (rng.randn(2, 5, 6, C) * 30).astype(np.float32)
    xt = nchw(x).double().requires_grad_()
    y = O.tf_lrn(xt, r, bias, alpha, beta)
    gy = rng.randn(2, 5, 6, C).astype(np.float32)
    (y * nchw(gy).double()).sum().backward()
    xb = torch.zeros(2, 5, 6, ld, device=DEV)
    xb[..., :C] = torch.tensor(x)
    gb = torch.zeros(2, 5, 6, ld, device=DEV)
    gb[..., :C] = torch.tensor(gy)
    yo, scale = ops.lrn_fwd(xb, C, r, bias, alpha, beta)
    assert rel(yo[..., :C], nhwc(y)) < 2e-6 and float(yo[..., C:].abs().sum()) == 0
    gx = ops.lrn_bwd(xb, yo, scale, gb, C, r, alpha, beta)
    assert rel(gx[..., :C], nhwc(xt.grad)) < 5e-6


# ---- the assembled network -------------------------------------------------------------------------------------------

def _net(upto, pool1=False, seed=11):
    from neural_flow_style_amd import inception
    w = inception.synthetic_weights(seed, upto=upto)
    return inception.InceptionV1(w, DEV, pool1=pool1), w


NAMES = ["conv2d0", "maxpool0", "localresponsenorm0", "conv2d1_pre_relu", "conv2d2", "localresponsenorm1", "maxpool1",
         "mixed3a", "mixed3a_3x3", "mixed3a_5x5_bottleneck", "mixed3a_pool", "mixed3b", "mixed3b_3x3_bottleneck_pre_relu",
         "maxpool4", "mixed4a", "mixed4a_pool_reduce_pre_relu", "mixed4b"]


@pytest.mark.parametrize("pool1,H,W", [(False, 45, 62), (True, 20, 24)])
def test_inception_forward_and_data_gradient_match_the_oracle(pool1, H, W):
    """every addressable tensor kind (ReLU outputs, *_pre_relu, pools, LRNs, module outputs incl. the 508-channel
    mixed4a, branch ranges) forward; one functional of all of them backward to the image"""
    net, w = _net("mixed4b", pool1)
    rng = np.random.RandomState(5)
    img = (rng.rand(2, H, W, 3) * 255).astype(np.float32)
    x = torch.tensor(img, device=DEV) - torch.tensor(O.VGG_MEAN, dtype=torch.float32, device=DEV)
    acts = net.forward(x.contiguous(), "mixed4b", keep=set(NAMES))
    xi = torch.tensor(img, dtype=torch.float64, requires_grad=True)
    w64 = {k: (np.asarray(a, np.float64), np.asarray(b, np.float64)) for k, (a, b) in w.items()}
    feats = O.inception_v1_features(xi, w64, "mixed4b", pool1=pool1)
    grads, total = {}, 0
    for i, n in enumerate(NAMES):
        c = acts.channels[n]
        assert c == feats[n].shape[-1] and acts[n].shape[-1] == (c + 63) // 64 * 64, n
        assert rel(acts[n][..., :c], feats[n]) < 2e-5, n
        assert float(acts[n][..., c:].abs().sum()) == 0, n
        r = torch.tensor(np.random.RandomState(i).randn(*feats[n].shape)) / feats[n].detach().abs().mean().clamp_min(1e-6)
        total = total + (feats[n] * r).sum()
        g = torch.zeros_like(acts[n])
        g[..., :c] = r.float().to(DEV)
        grads[n] = g
    total.backward()
    g_x = net.backward(acts, grads, "mixed4b")
    assert g_x.shape == (2, H, W, 3)
    assert rel(g_x, xi.grad) < 5e-5


def test_inception_widths_come_from_the_weights_and_bad_sets_are_refused(tmp_path):
    from neural_flow_style_amd import inception
    w = inception.synthetic_weights(3, upto="mixed3a")
    # a file with other widths than the published table runs: nothing is hard-wired
    rng = np.random.RandomState(0)
    w["mixed3a_1x1"] = ((rng.randn(1, 1, 192, 40) * 0.1).astype(np.float32), np.zeros(40, np.float32))
    net = inception.InceptionV1(w, DEV)
    assert net.cout["mixed3a"] == 40 + 128 + 32 + 32
    acts = net.forward(torch.randn(1, 16, 16, 3, device=DEV), "mixed3a")
    assert acts["mixed3a"].shape == (1, 2, 2, 256) and acts.channels["mixed3a"] == 232
    bad = dict(w)
    bad["conv2d2"] = (np.zeros((3, 3, 32, 192), np.float32), np.zeros(192, np.float32))
    with pytest.raises(ValueError, match="conv2d2"):
        inception.InceptionV1(bad, DEV)
    # the .npz loader: keys of the graph's Const nodes, LRN attributes optional; missing units raise
    full = inception.synthetic_weights(3, upto="mixed3b")
    path = tmp_path / "tensorflow_inception_graph.npz"
    arrays = {}
    for k, (a, b) in full.items():
        arrays[k + "_w"], arrays[k + "_b"] = a, b
    np.savez(path, localresponsenorm0=np.array([2, 1.0, 2e-5, 0.75]), **arrays)
    with pytest.raises(KeyError, match="mixed4a_1x1"):
        inception.load_inception(str(tmp_path / "tensorflow_inception_graph.pb"), DEV)
    got, lrn = inception.load_npz_weights(str(path), upto="mixed3b_pool_reduce")
    assert list(got) == list(full) and lrn == {"localresponsenorm0": (2, 1.0, 2e-5, 0.75)}
    with pytest.raises(FileNotFoundError, match="SYNTHETIC"):
        inception.load_inception(str(tmp_path / "nothing" / "tensorflow_inception_graph.pb"), DEV, synthetic=False)


# ---- through the engine ----------------------------------------------------------------------------------------------

def _grid_case(G=28, V=2):
    from neural_flow_style_amd import synthetic as S
    rng = np.random.RandomState(21)
    d0 = S.blob_density(G, rng)
    vel0 = (rng.randn(G, G, G, 3) * 0.3 / (G - 1)).astype(np.float32)
    simg = S.style_image(int(G * 1.5), int(G * 1.5), rng)
    return d0, vel0, simg, S.uniform_views(V)


@pytest.mark.parametrize("content", [None, ("mixed3b_3x3_bottleneck_pre_relu", 44), ("mixed4b_pool_reduce_pre_relu", 16),
                                     ("mixed3a", 0)])
def test_engine_gradient_parity_with_the_inception_style_layers_of_the_reference_driver(content):
    """test_smokegun.py:140-141: network tensorflow_inception_graph.pb, style_layer ['conv2d2','mixed3b','mixed4b']
    (480- and 512-channel module outputs, the first in padded rows), resize_scale 1.5; with run.bat:14-20's content
    terms on '*_pre_relu' tensors (channel maximisation, |.| of a signed tensor) and -mean on a module output"""
    from neural_flow_style_amd import engine, transform as T
    net, w = _net("mixed4b", seed=123)
    d0, vel0, simg, mats = _grid_case()
    layers = ["conv2d2", "mixed3b", "mixed4b"]
    kw = {}
    if content:
        kw = dict(w_content=3e3, content_layer=content[0], content_channel=content[1])
    loss = engine.RenderStyleLoss(net, layers, [1.0, 0.5, 2.0], 1.0, transmit=0.05, resize_scale=1.5, **kw)
    loss.set_style_image(simg)
    gs = engine.GridStylizer(loss, torch.tensor(d0, device=DEV), k=3, target="v", lr=1e-3)
    gs.var.copy_(torch.tensor(vel0))
    losses, g_h = gs.gradient(T.rot_to_device(mats, DEV))

    cfg = dict(k=3, transmit=0.05, style_layer=layers, w_style_layer=[1.0, 0.5, 2.0], w_style=1.0, upto="mixed4b",
               network="tensorflow_inception_graph.pb", resize_scale=1.5)
    if content:
        cfg.update(w_content=3e3, content_layer=content[0], content_channel=content[1])
    sfe = O.style_target_features(torch.tensor(simg)[None], w, layers, upto="mixed4b", cfg=cfg)
    v = torch.tensor(vel0)[None].requires_grad_()
    total, per_view, _ = O.grid_forward(torch.tensor(d0)[None, ..., None], v, torch.tensor(np.asarray(mats, np.float32)),
                                        cfg, w, sfe)
    total.backward()
    if content:
        cfg0 = dict(cfg, w_content=0)
        t0, _, _ = O.grid_forward(torch.tensor(d0)[None, ..., None], torch.tensor(vel0)[None],
                                  torch.tensor(np.asarray(mats, np.float32)), cfg0, w, sfe)
        assert abs(float(total.detach()) - float(t0.detach())) > 1e-3 * abs(float(t0.detach()))     # the content term is not negligible
    np.testing.assert_allclose(losses.cpu().numpy(), [float(l.detach()) for l in per_view], rtol=2e-4)
    assert rel(g_h, v.grad[0]) < 1e-3


# ---- the Styler surface ----------------------------------------------------------------------------------------------

@pytest.mark.parametrize("w_style,content", [(1.0, ("mixed3b_3x3_bottleneck_pre_relu", 44)), (0.0, ("mixed3a_3x3_bottleneck_pre_relu", 65))])
def test_styler3p_with_the_inception_network_matches_the_oracle_loop(w_style, content):
    """the reference driver's own configuration (test_smokegun.py:128-148): density field, one unrotated view, network
    tensorflow_inception_graph.pb, style layers conv2d2 / mixed3b / mixed4b, resize_scale 1.5; content term of
    run.bat:14 (channel 44 of mixed3b_3x3_bottleneck_pre_relu) -- with the style term, and alone (w_style 0, no style
    image: the semantic-transfer runs)"""
    from neural_flow_style_amd import inception, synthetic as S
    from neural_flow_style_amd.styler_3p import Styler
    from tests.test_styler_gpu import _config, _particles
    G, n, nk, F = 24, 3000, 2, 1
    rng = np.random.RandomState(9)
    frames = [_particles(G, n, nk, rng) for _ in range(F)]
    simg = S.style_image(36, 36, rng)
    layers = ["conv2d2", "mixed3b", "mixed4b"] if w_style else ["conv2d2"]
    cfg = _config(network="tensorflow_inception_graph.pb", resolution=[G, G, G], domain=[G, G, G], radius=0.5, nsize=1,
                  support=4, rest_density=1000, k=3, clip=False, target_field="d", num_frames=F, batch_size=1,
                  frames_per_opt=1, window_sigma=1.0, interp=1, lr=0.05, iter=3, octave_n=1, octave_scale=1.8,
                  style_layer=layers, w_style_layer=[1] * len(layers), w_style=w_style, w_content=50.0,
                  content_layer=content[0], content_channel=content[1], transmit=0.1, rotate=False, v_batch=1,
                  resize_scale=1.5, views_mode="sequential", style_target=simg if w_style else "", num_kernels=nk,
                  kernel_scale=2, w_pressure=0, w_density=0)
    st = Styler(cfg)
    assert "synthetic" in st.net.source and type(st.net).__name__ == "InceptionV1"
    st.load_img([G, G])
    params = {"p": [f[0] for f in frames], "r": [f[1] for f in frames]}
    res = st.run(params)
    ocfg = dict(vars(cfg))
    w = inception.synthetic_weights(cfg.seed)
    hist, g_opt, d_fin = O.styler3p_run(ocfg, params, w, [simg], None, views_mode="sequential")
    np.testing.assert_allclose(res["l"][0], hist[0], rtol=2e-3)
    assert rel(res["opt"][0], g_opt[0]) < 2e-3
    assert rel(res["d"][0], d_fin[0]) < 1e-3
    if w_style:
        hist0, _, _ = O.styler3p_run(dict(ocfg, w_content=0), params, w, [simg], None, views_mode="sequential")
        assert abs(hist[0][0] - hist0[0][0]) > 1e-3 * abs(hist0[0][0])       # the content term is not negligible


def test_smokegun_driver_follows_the_reference_override_block(tmp_path, monkeypatch):
    """main() of test_smokegun.py:111-197 value for value when no flag is given (Inception graph, conv2d2 / mixed3b /
    mixed4b, one unrotated view, resize_scale 1.5); BASELINE's VGG configuration through flags; a short demo run on
    the Inception network"""
    import os
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import test_smokegun as drv
    from config import get_config

    def cfg_for(argv):
        monkeypatch.setattr(sys, "argv", ["test_smokegun.py"] + argv)
        cfg, _ = get_config()
        cfg.log_dir, cfg.data_dir = str(tmp_path / "log"), str(tmp_path / "nodata")
        return cfg

    seen = {}
    monkeypatch.setattr(drv, "run", lambda c: seen.update(vars(c)))
    drv.main(cfg_for([]))
    assert seen["network"] == "tensorflow_inception_graph.pb" and seen["style_layer"] == ["conv2d2", "mixed3b", "mixed4b"]
    assert seen["w_style_layer"] == [1, 1, 1] and seen["rotate"] is False and seen["resolution"] == [200, 300, 200]
    assert seen["resize_scale"] == 1.5 and seen["transmit"] == 0.01 and seen["iter"] == 20 and seen["lr"] == 0.1
    assert seen["num_kernels"] == 2 and seen["target_field"] == "d" and seen["radius"] == 0.5 and seen["k"] == 3
    assert seen["content_layer"] == "mixed4d_3x3_bottleneck_pre_relu" and seen["content_channel"] == 139
    assert seen["w_content"] == 1 and seen["w_style"] == 0
    seen.clear()
    drv.main(cfg_for(["--network", "vgg_19.ckpt", "--rotate", "true", "--n_views", "8", "--w_style", "1"]))
    assert seen["network"] == "vgg_19.ckpt" and seen["style_layer"] == ["conv1_1", "conv2_1", "conv3_1", "conv4_1", "conv5_1"]
    assert seen["rotate"] is True and seen["n_views"] == 8 and seen["w_content"] == 0
    monkeypatch.undo()

    argv = ["--resolution", "24", "36", "24", "--iter", "2", "--synthetic_weights", "true", "--content_layer",
            "mixed3b_3x3_bottleneck_pre_relu", "--content_channel", "44", "--target_frame", "70"]
    monkeypatch.setattr(sys, "argv", ["test_smokegun.py"] + argv)
    cfg, _ = get_config()
    cfg.log_dir, cfg.data_dir = str(tmp_path / "log2"), str(tmp_path / "nodata")
    res = drv.main(cfg)                                   # the reference's default: semantic transfer, no style image
    assert res["d"][0].shape[:3] == (24, 36, 24) and len(res["l"][0]) == 2 and np.isfinite(res["l"][0]).all()
    assert os.path.exists(os.path.join(cfg.log_dir, "070.png")) and os.path.exists(os.path.join(cfg.log_dir, "070.npz"))


def test_grid_stylizer_steps_with_the_inception_network_replayed_as_a_hipgraph():
    """small grids replay forward + adjoint as one hipGraph (GridStylizer decides by problem size): the Inception
    network's grouped launches, zero fills and channel-range copies must capture and replay to the eager trajectory"""
    from neural_flow_style_amd import engine, transform as T
    net, _ = _net("mixed4b", seed=123)
    d0, vel0, simg, mats = _grid_case()
    layers = ["conv2d2", "mixed3b", "mixed4b"]
    rot = T.rot_to_device(mats, DEV)
    hist = {}
    for graph in (False, True):
        loss = engine.RenderStyleLoss(net, layers, [1.0, 0.5, 2.0], 1.0, transmit=0.05, resize_scale=1.5,
                                      w_content=3e3, content_layer="mixed3b_3x3_bottleneck_pre_relu", content_channel=44)
        loss.set_style_image(simg)
        gs = engine.GridStylizer(loss, torch.tensor(d0, device=DEV), k=3, target="v", lr=2e-3, graph=graph)
        gs.var.copy_(torch.tensor(vel0))
        hist[graph] = ([float(gs.step(rot)) for _ in range(5)], gs.var.clone())
        assert bool(gs.use_graph) == graph
    np.testing.assert_allclose(hist[True][0], hist[False][0], rtol=1e-5)
    assert rel(hist[True][1], hist[False][1]) < 1e-5
    assert hist[False][0][-1] < hist[False][0][0]


@pytest.mark.parametrize("style_mask", [False, True])
def test_image_style_loss_2d_with_the_inception_network(style_mask):
    """the 2-D colour path (styler_2p.py:91-102) on the Inception graph: Gram style loss on conv2d2 / mixed3a / the
    480-channel mixed3b (padded rows), TV, optionally the density mask on the features (styler_base.py:165-169)"""
    from neural_flow_style_amd import engine, synthetic as S
    net, w = _net("mixed3b", seed=5)
    rng = np.random.RandomState(2)
    H, W = 40, 56
    layers, wl = ["conv2d2", "mixed3a", "mixed3b"], [1.0, 2.0, 0.5]
    simg = S.style_image(H, W, rng)
    d = rng.rand(1, H, W, 3).astype(np.float32)
    dg = (rng.rand(1, H, W, 1) > 0.3).astype(np.float32)
    il = engine.ImageStyleLoss(net, layers, wl, 1.0, w_tv=0.01, style_mask=style_mask)
    il.set_style_image(simg)
    loss, g = il.loss_and_grad(torch.tensor(d, device=DEV), torch.tensor(dg, device=DEV) if style_mask else None)

    cfg = dict(network="tensorflow_inception_graph.pb")
    dt = torch.tensor(d).requires_grad_()
    d_img = O.plugin_to_loss_net(dt, 1.0, is_color=True)
    feats = O.loss_net_features(d_img, w, "mixed3b", cfg)
    sfe = O.style_target_features(torch.tensor(simg)[None], w, layers, upto="mixed3b", cfg=cfg)
    ls, _ = O.style_loss(feats, sfe, layers, wl, 1.0, d_gray=torch.tensor(dg) if style_mask else None)
    total = ls + 0.01 * O.tv_loss(d_img)
    total.backward()
    assert abs(float(loss.sum()) - float(total.detach())) < 2e-4 * abs(float(total.detach()))
    assert rel(g, dt.grad) < 1e-3


# ---- the main classifier: avgpool0 -> softmax2_pre_activation (styler_base.py:240-245) ---------------------------------

@pytest.mark.parametrize("B,H,W,C,k", [(2, 9, 11, 64, 7), (1, 7, 7, 1024, 7), (1, 10, 15, 128, 3)])
def test_avgpool_valid_and_its_adjoint(B, H, W, C, k):
    from neural_flow_style_amd import ops
    rng = np.random.RandomState(H)
    x = rng.randn(B, H, W, C).astype(np.float32)
    xt = nchw(x).double().requires_grad_()
    y = torch.nn.functional.avg_pool2d(xt, k, 1)
    gy = rng.randn(*nhwc(y).shape).astype(np.float32)
    (y * nchw(gy).double()).sum().backward()
    got = ops.avgpool_valid_fwd(torch.tensor(x, device=DEV), k)
    assert rel(got, nhwc(y)) < 1e-6
    gx = ops.avgpool_valid_bwd(torch.tensor(gy, device=DEV), (H, W), k)
    assert rel(gx, nhwc(xt.grad)) < 1e-6
    base = torch.full((B, H, W, C), 0.25, device=DEV)
    assert rel(ops.avgpool_valid_bwd(torch.tensor(gy, device=DEV), (H, W), k, gx=base) - 0.25, nhwc(xt.grad)) < 1e-6


def test_inception_classifier_logits_and_their_data_gradient():
    """the whole graph down to 'softmax2_pre_activation' at 224 x 256 (a 7 x 8 map in front of the 7 x 7 pool: two rows
    of logits per image, as the graph's reshape to [-1, 1024] gives them) against the oracle, and its data gradient"""
    net, w = _net("softmax2_pre_activation", seed=4)
    rng = np.random.RandomState(8)
    img = (rng.rand(1, 224, 256, 3) * 255).astype(np.float32)
    x = torch.tensor(img, device=DEV) - torch.tensor(O.VGG_MEAN, dtype=torch.float32, device=DEV)
    names = ["mixed4e", "mixed5b", "avgpool0", "softmax2_pre_activation"]
    acts = net.forward(x.contiguous(), "softmax2_pre_activation", keep=set(names))
    # (float64 oracle: in float32 the restatement itself routes a few max-pool near-ties differently, 7e-3 on the gradient)
    xi = torch.tensor(img, dtype=torch.float64, requires_grad=True)
    w64 = {k: (np.asarray(a, np.float64), np.asarray(b, np.float64)) for k, (a, b) in w.items()}
    feats = O.inception_v1_features(xi, w64, "softmax2_pre_activation")
    assert tuple(feats["softmax2_pre_activation"].shape) == (1, 1, 2, 1008)
    total, grads = 0, {}
    for i, n in enumerate(names):
        c = acts.channels[n]
        assert rel(acts[n][..., :c], feats[n]) < 5e-5, n
        r = torch.tensor(np.random.RandomState(i).randn(*feats[n].shape)) / feats[n].detach().abs().mean()
        total = total + (feats[n] * r).sum()
        g = torch.zeros_like(acts[n])
        g[..., :c] = r.float().to(DEV)
        grads[n] = g
    total.backward()
    assert rel(net.backward(acts, grads, "softmax2_pre_activation"), xi.grad) < 5e-5
    small = torch.zeros(1, 100, 100, 3, device=DEV)
    with pytest.raises(ValueError, match="avgpool0"):
        net.forward(small, "softmax2_pre_activation")


def test_engine_gradient_with_the_top_k_content_target_on_the_classifier_logits():
    """styler_base.py:232-247 with the flag defaults' mechanism: a content image, content_layer
    softmax2_pre_activation, top_k 5 -- the target keeps the five strongest logits of the content image, the term is
    mean((logits - amp * target)^2).  Through the 2-D colour loss on a textured 224 x 256 image: a smooth render has
    large areas whose neighbouring pixels differ by less than a float32 ulp of the mean-subtracted input -- exact ties
    in the graph's 16 max pools in float32, distinct values in a float64 oracle, i.e. the same loss with its gradient
    routed to other pixels (measured: 1.4e-2 on a blob, 6e-3 on a noise volume; the subgradient of a max is not unique
    there, in TF either; even on a textured image ONE near-tie among the ~10^6 pool windows moves 2e-3 of the gradient:
    weight seed 123 has one for this image, seed 4 has none)"""
    from neural_flow_style_amd import engine, synthetic as S
    net, w = _net("softmax2_pre_activation", seed=4)
    rng = np.random.RandomState(12)
    H, W = 224, 256
    d = rng.rand(1, H, W, 3).astype(np.float32)
    cimg = S.style_image(H, W, rng)
    il = engine.ImageStyleLoss(net, ["conv2d2"], [1.0], 0.0, w_content=1.0, content_layer="softmax2_pre_activation",
                               w_content_amp=100.0)
    cf = il.set_content_image(cimg, top_k=5)
    assert tuple(cf.shape) == (1, 1, 2, 1008) and int((cf != 0).sum()) == 10
    loss, g = il.loss_and_grad(torch.tensor(d, device=DEV))

    cfg = dict(network="tensorflow_inception_graph.pb")
    w64 = {k: (np.asarray(a, np.float64), np.asarray(b, np.float64)) for k, (a, b) in w.items()}
    target = O.content_target_feature(torch.tensor(cimg, dtype=torch.float64)[None], w64, "softmax2_pre_activation", cfg,
                                      top_k=5)
    assert rel(cf, target) < 1e-5
    dt = torch.tensor(d, dtype=torch.float64).requires_grad_()
    feats = O.loss_net_features(O.plugin_to_loss_net(dt, 1.0, is_color=True), w64, "softmax2_pre_activation", cfg)
    total = O.content_loss(feats["softmax2_pre_activation"], 0, target, 100.0)
    total.backward()
    assert abs(float(loss.sum()) - float(total.detach())) < 1e-5 * abs(float(total.detach()))
    assert rel(g, dt.grad) < 1e-4
    with pytest.raises(AssertionError, match="softmax2_pre_activation"):
        engine.ImageStyleLoss(net, ["conv2d2"], [1.0], 0.0, w_content=1.0, content_layer="mixed3b").set_content_image(
            cimg, top_k=5)
