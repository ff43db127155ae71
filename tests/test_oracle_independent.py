"""oracle (PyTorch, nfs_oracle.py) == second restatement (float64 NumPy loops + hand-derived adjoints,
np_restatement.py), values and gradients, for the ops whose parity the reference itself cannot pin (no TF-1.15
here): VGG conv / avg-pool chain, Gram + style loss, render with its global max, legacy resizes, ApplyAdam,
smoothing conv.  Two implementations written separately from the cited lines: a shared misreading would show."""
import numpy as np
import torch

from oracle import nfs_oracle as O
from oracle import np_restatement as R

DT = torch.float64


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


def _weights(upto, width_div=8, scale=1.0, bias_scale=1.0, seed=123):
    w = O.synthetic_vgg19_weights(seed, upto=upto, dtype=np.float64, width_div=width_div)
    return {k: (v[0] * scale, v[1] * bias_scale) for k, v in w.items()}


def test_vgg_conv_pool_chain_and_style_gradient():
    rng = np.random.RandomState(0)
    layers = ["conv1_1", "conv2_1", "conv3_1"]
    w = _weights("conv3_1")
    d_img = rng.uniform(0, 255, (1, 13, 10, 3))                      # odd size: the VALID pools drop a row
    s_img = rng.uniform(0, 255, (1, 13, 10, 3))
    x = torch.tensor(d_img, dtype=DT, requires_grad=True)
    fo = O.vgg19_features(x, w, "conv3_1")
    so = O.style_target_features(torch.tensor(s_img, dtype=DT), w, layers, upto="conv3_1")
    lo, _ = O.style_loss(fo, so, layers, [1.0, 0.5, 2.0], w_style=0.7)
    (go,) = torch.autograd.grad(lo, x)

    fr, tape = R.vgg19_forward(d_img, w, "conv3_1")
    sr, _ = R.vgg19_forward(s_img, w, "conv3_1")
    for name in ("conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1"):
        assert fr[name].shape == tuple(fo[name].shape)
        assert rel(fr[name], fo[name].detach().numpy()) < 1e-12, name
    assert fr["conv2_1"].shape[1:3] == (6, 5) and fr["conv3_1"].shape[1:3] == (3, 2)   # floor(13/2), floor(6/2)
    lr, gr = R.style_loss_and_grad(fr, sr, layers, [1.0, 0.5, 2.0], w_style=0.7)
    assert abs(lr - float(lo)) < 1e-12 * abs(float(lo))
    g_img = R.vgg19_backward(fr, tape, w, gr)
    assert rel(g_img, go.numpy()) < 1e-11


def test_vgg_chain_at_real_vgg_dynamic_range():
    """real VGG-19 has conv1_1 weights of O(0.5) on 0..255 inputs and activations of O(10^2..10^3): the same chain
    with the first-layer weights scaled up so that the activations reach that range"""
    rng = np.random.RandomState(1)
    w = _weights("conv2_1")
    w["conv1_1"] = (w["conv1_1"][0] * 12.0, w["conv1_1"][1] * 100.0)
    d_img = rng.uniform(0, 255, (1, 8, 8, 3))
    x = torch.tensor(d_img, dtype=DT)
    fo = O.vgg19_features(x, w, "conv2_1")
    fr, _ = R.vgg19_forward(d_img, w, "conv2_1")
    assert float(fo["conv1_1"].max()) > 300
    for name in fr:
        assert rel(fr[name], fo[name].numpy()) < 1e-12


def test_gram_matrix_loops():
    rng = np.random.RandomState(2)
    x = rng.randn(2, 5, 4, 6)
    go = O.gram_matrix(torch.tensor(x, dtype=DT))
    for b in range(2):
        assert rel(R.gram(x[b]), go[b].numpy()) < 1e-13


def test_render_and_global_max_gradient_including_ties():
    rng = np.random.RandomState(3)
    for liquid in (False, True):
        d = rng.uniform(0, 1, (2, 7, 5, 4))
        d[d < 0.3] = 0.0
        if not liquid:
            d[1] = d[0]                                              # the global maximum is attained twice
        g = rng.randn(2, 5, 4)
        dt = torch.tensor(d[..., None], dtype=DT, requires_grad=True)
        img = O.render(dt, 0.37, liquid)
        (go,) = torch.autograd.grad((img[..., 0] * torch.tensor(g, dtype=DT)).sum(), dt)
        assert rel(R.render(d, 0.37, liquid), img[..., 0].detach().numpy()) < 1e-13
        assert rel(R.render_bwd(d, 0.37, g, liquid), go[..., 0].numpy()) < 1e-12


def test_legacy_resizes():
    rng = np.random.RandomState(4)
    x = rng.rand(1, 8, 6, 2)
    for oh, ow in ((12, 9), (4, 3), (8, 6), (16, 6)):
        a = O.tf1_resize_bilinear(torch.tensor(x, dtype=DT), oh, ow).numpy()
        assert np.abs(R.tf1_resize_bilinear(x, oh, ow) - a).max() < 1e-6, (oh, ow)
        b = O.tf1_resize_bicubic(torch.tensor(x, dtype=DT), oh, ow).numpy()
        assert np.abs(R.tf1_resize_bicubic(x, oh, ow) - b).max() < 1e-6, (oh, ow)
    # the two classic checkpoints of the legacy kernels: identity at equal size, pixel replication weights at 2x
    assert np.abs(R.tf1_resize_bilinear(x, 8, 6) - x).max() == 0
    up = R.tf1_resize_bilinear(x, 16, 12)
    assert np.abs(up[:, ::2, ::2] - x).max() < 1e-12
    assert np.abs(up[:, 1, 0] - 0.5 * (x[:, 0, 0] + x[:, 1, 0])).max() < 1e-7


def test_tf_adam_trajectory():
    rng = np.random.RandomState(5)
    x0 = rng.randn(50)
    grads = [rng.randn(50) * s for s in (1.0, 1e-7, 3.0, 1e-3, 1e-9)]  # incl. magnitudes where eps placement matters
    want = R.adam_tf_trajectory(x0, grads, 0.1)
    opt = O.TFAdam()
    x = torch.tensor(x0, dtype=DT)
    for t, g in enumerate(grads):
        x = opt.step(x, torch.tensor(g, dtype=DT), 0.1)
        assert rel(x.numpy(), want[t]) < 1e-13
    # and it is NOT torch.optim.Adam (epsilon placement): at gradient magnitudes near eps the trajectories separate
    tiny = [rng.randn(50) * 1e-7 for _ in range(5)]
    want_t = R.adam_tf_trajectory(x0, tiny, 0.1)
    opt_t = O.TFAdam()
    x = torch.tensor(x0, dtype=DT)
    for g in tiny:
        x = opt_t.step(x, torch.tensor(g, dtype=DT), 0.1)
    assert rel(x.numpy(), want_t[-1]) < 1e-13
    p = torch.nn.Parameter(torch.tensor(x0, dtype=DT))
    ta = torch.optim.Adam([p], lr=0.1)
    for g in tiny:
        p.grad = torch.tensor(g, dtype=DT)
        ta.step()
    assert rel(p.detach().numpy() - x0, want_t[-1] - x0) > 0.3


def test_smoothing_conv():
    rng = np.random.RandomState(6)
    d = rng.randn(6, 5, 7)
    a = O.smooth3d_relu(torch.tensor(d, dtype=DT)[None, ..., None], 3)[0, ..., 0].numpy()
    assert rel(R.smooth3d_relu(d, 3), np.abs(a)) < 1e-13            # abs: the oracle keeps -0.0 for the TF tie mask
