"""Inception-v1 loss network, host side and oracle (no GPU): the TF op semantics the oracle restates for it (SAME
padding split, max pool over in-range taps, tf.nn.lrn) against hand-computed answers; the published tensor shapes of
the "inception5h" graph at its native 224 x 224 input; the weight-file loader (keys of the graph's Const nodes,
strictness, explicit opt-in for synthetic weights).  styler_base.py:17-30, 51-57, 91-94."""
import numpy as np
import pytest
import torch

from neural_flow_style_amd import inception
from oracle import nfs_oracle as O


def test_tf_same_padding_puts_the_odd_pad_after():
    """TF SAME: out = ceil(n / s), total = max((out-1) s + k - n, 0), before = total // 2 -- with an odd total the extra
    zero goes AFTER (PyTorch's symmetric `padding=` cannot say this)"""
    x = torch.arange(1.0, 5.0).view(1, 1, 1, 4)                        # 1 2 3 4
    w = np.ones((1, 3, 1, 1), np.float32)
    y = O.tf_conv2d_same(x, w, stride=2)                               # out 2, total 1 -> before 0, after 1
    assert y.flatten().tolist() == [1 + 2 + 3, 3 + 4 + 0]
    x5 = torch.arange(1.0, 6.0).view(1, 1, 1, 5)                       # out 3, total 2 -> 1 / 1
    assert O.tf_conv2d_same(x5, w, stride=2).flatten().tolist() == [0 + 1 + 2, 2 + 3 + 4, 4 + 5 + 0]
    assert O.tf_conv2d_same(x, w, stride=1).flatten().tolist() == [3.0, 6.0, 9.0, 7.0]
    from neural_flow_style_amd import ops
    assert ops.same_out(4, 3, 2) == (2, 0) and ops.same_out(5, 3, 2) == (3, 1) and ops.same_out(300, 7, 2) == (150, 2)
    assert ops.same_out(7, 5, 1) == (7, 2) and ops.same_out(1, 3, 2) == (1, 1)


def test_tf_maxpool_ignores_the_padding_and_lrn_formula():
    x = -torch.arange(1.0, 5.0).view(1, 1, 1, 4)                       # all negative: zero padding would win
    assert O.tf_maxpool3_same(x.expand(1, 1, 3, 4), 2)[0, 0, 0].tolist() == [-1.0, -3.0]
    assert O.tf_maxpool3_same(x.expand(1, 1, 3, 4), 1)[0, 0, 1].tolist() == [-1.0, -1.0, -2.0, -3.0]
    # tf.nn.lrn, depth_radius 1: channel c sees c-1 .. c+1
    v = torch.tensor([1.0, 2.0, 3.0]).view(1, 3, 1, 1)
    y = O.tf_lrn(v, 1, 2.0, 0.5, 0.75).flatten()
    want = [1 / (2 + 0.5 * (1 + 4)) ** 0.75, 2 / (2 + 0.5 * (1 + 4 + 9)) ** 0.75, 3 / (2 + 0.5 * (4 + 9)) ** 0.75]
    np.testing.assert_allclose(y.numpy(), want, rtol=1e-6)


def test_published_tensor_shapes_of_the_graph_at_its_native_input():
    """the 5h graph at 224 x 224: conv2d0 112^2 x 64, conv2d2 56^2 x 192, mixed3a 28^2 x 256, mixed3b 480, mixed4a
    14^2 x 508, mixed4b .. 4c 512, mixed4d 528, mixed4e 832, mixed5a 7^2 x 832, mixed5b 1024; run.bat's content tensors
    exist with room for their channel numbers (run.bat:14-20)"""
    w = inception.synthetic_weights(0)
    assert len(w) == len(inception.conv_units()) + 1 == 3 + 9 * 6 + 1 and w["softmax2"][0].shape == (1, 1, 1024, 1008)
    feats = O.inception_v1_features(torch.zeros(1, 224, 224, 3), w, "softmax2_pre_activation")
    assert tuple(feats["avgpool0"].shape) == (1, 1, 1, 1024) and tuple(feats["softmax2_pre_activation"].shape) == (1, 1, 1, 1008)
    want = {"conv2d0": (112, 64), "maxpool0": (56, 64), "conv2d1": (56, 64), "conv2d2": (56, 192), "maxpool1": (28, 192),
            "mixed3a": (28, 256), "mixed3b": (28, 480), "maxpool4": (14, 480), "mixed4a": (14, 508), "mixed4b": (14, 512),
            "mixed4c": (14, 512), "mixed4d": (14, 528), "mixed4e": (14, 832), "maxpool10": (7, 832), "mixed5a": (7, 832),
            "mixed5b": (7, 1024)}
    for name, (hw, c) in want.items():
        assert tuple(feats[name].shape) == (1, hw, hw, c), name
    assert feats["mixed3b_3x3_bottleneck_pre_relu"].shape[-1] > 65           # run.bat: channels 44, 65
    assert feats["mixed4b_pool_reduce_pre_relu"].shape[-1] > 60              # run.bat: channels 6, 16, 38, 60
    assert feats["mixed4d_3x3_bottleneck_pre_relu"].shape[-1] > 139          # config.py:77: the flag default, channel 139
    assert inception.unit_of("mixed4b_pool_reduce_pre_relu") == "mixed4b" and inception.unit_of("maxpool4") == "maxpool4"
    with pytest.raises(KeyError):
        inception.unit_of("conv3_1")
    assert O.inception_last_layer(["mixed4b", "conv2d2", "mixed3b_5x5"]) == "mixed4b"


def _save(path, w, **extra):
    arrays = {}
    for k, (a, b) in w.items():
        arrays[k + "_w"], arrays[k + "_b"] = a, b
    arrays.update(extra)
    np.savez(path, **arrays)


def test_weight_file_loader_is_strict(tmp_path):
    w = inception.synthetic_weights(1, upto="mixed3a")
    small = {k: (a[..., :4, :8] if a.shape[2] > 4 else a[..., :8], b[:8]) for k, (a, b) in w.items()}   # (light file)
    path = str(tmp_path / "tensorflow_inception_graph.npz")
    _save(path, small, localresponsenorm1=np.array([4, 1.0, 1e-3, 0.75]))
    got, lrn = inception.load_npz_weights(path, upto="mixed3a")
    assert list(got) == list(small) and lrn == {"localresponsenorm1": (4, 1.0, 1e-3, 0.75)}
    for k in got:
        np.testing.assert_array_equal(got[k][0], small[k][0])
    with pytest.raises(KeyError, match="mixed3b_1x1"):                 # the whole graph is asked for by default
        inception.load_npz_weights(path)
    bad = dict(small)
    bad["conv2d1"] = (small["conv2d1"][0][0], small["conv2d1"][1])      # not 4-D
    _save(path, bad)
    with pytest.raises(ValueError, match="conv2d1"):
        inception.load_npz_weights(path, upto="mixed3a")
    _save(path, small, localresponsenorm0=np.array([5, 2.0]))
    with pytest.raises(ValueError, match="localresponsenorm0"):
        inception.load_npz_weights(path, upto="mixed3a")


def test_loader_raises_without_weights_unless_synthetic_is_asked_for(tmp_path, monkeypatch):
    monkeypatch.delenv("NFS_SYNTHETIC_VGG", raising=False)
    pb = str(tmp_path / "tensorflow_inception_graph.pb")
    with pytest.raises(FileNotFoundError, match="SYNTHETIC"):
        inception.load_inception(pb, "cpu")
    open(pb, "wb").write(b"\x0a\x00")
    with pytest.raises(FileNotFoundError, match="cannot be parsed without TensorFlow"):
        inception.load_inception(pb, "cpu")
