"""vgg.load_npz_weights / load_vgg (counterpart of vgg.py:110-120): key forms, strictness, explicit opt-in for
synthetic weights.  Host logic only (no GPU): the VGG object itself is not constructed here."""
import numpy as np
import pytest

from neural_flow_style_amd import vgg


def _write(path, form, blocks=vgg.VGG19_BLOCKS, drop=None, scope="vgg_19", bad_shape=None):
    rng = np.random.RandomState(0)
    out = {}
    want = {}
    for name, kind, cin, cout in vgg.layer_sequence(blocks):
        if kind != "conv" or name == drop:
            continue
        w = rng.randn(3, 3, cin, cout).astype(np.float32) if cin * cout <= 64 * 128 else \
            np.full((3, 3, cin, cout), float(len(out)), np.float32)
        if name == bad_shape:
            w = w[:2]
        b = rng.randn(cout).astype(np.float32)
        wk, bk = {"short": ("%s/weights" % name, "%s/biases" % name),
                  "ckpt": ("%s/%s/%s/weights" % (scope, name[:5], name), "%s/%s/%s/biases" % (scope, name[:5], name)),
                  "wb": (name + "_w", name + "_b")}[form]
        out[wk], out[bk] = w, b
        want[name] = (w, b)
    np.savez(path, **out)
    return want


@pytest.mark.parametrize("form", ["short", "ckpt", "wb"])
def test_load_npz_weights_accepts_the_three_key_forms(tmp_path, form):
    path = str(tmp_path / "vgg_19.npz")
    want = _write(path, form)
    got = vgg.load_npz_weights(path)
    assert list(got) == [n for n, k, _, _ in vgg.layer_sequence() if k == "conv"] and len(got) == 16
    for name in ("conv1_1", "conv3_4", "conv5_4"):
        assert np.array_equal(got[name][0], want[name][0]) and np.array_equal(got[name][1], want[name][1])
        assert got[name][0].dtype == np.float32


def test_vgg16_checkpoint_scope(tmp_path):
    path = str(tmp_path / "vgg_16.npz")
    _write(path, "ckpt", vgg.VGG16_BLOCKS, scope="vgg_16")
    got = vgg.load_npz_weights(path, vgg.VGG16_BLOCKS, scope="vgg_16")
    assert len(got) == 13 and "conv3_4" not in got
    with pytest.raises(KeyError):
        vgg.load_npz_weights(path, vgg.VGG16_BLOCKS, scope="vgg_19")    # wrong scope prefix must not match silently


def test_missing_or_misshaped_layer_raises(tmp_path):
    path = str(tmp_path / "vgg_19.npz")
    _write(path, "short", drop="conv3_2")
    with pytest.raises(KeyError, match="conv3_2"):
        vgg.load_npz_weights(path)
    assert list(vgg.load_npz_weights(path, upto="conv3_1"))[-1] == "conv3_1"   # layers above ``upto`` are not needed
    _write(path, "short", bad_shape="conv2_1")
    with pytest.raises(ValueError, match="conv2_1"):
        vgg.load_npz_weights(path)


def test_load_vgg_without_weights_raises_unless_synthetic_is_requested(tmp_path, monkeypatch):
    monkeypatch.delenv("NFS_SYNTHETIC_VGG", raising=False)
    with pytest.raises(FileNotFoundError, match="synthetic"):
        vgg.load_vgg(str(tmp_path / "vgg_19.ckpt"), "cpu")
    (tmp_path / "vgg_19.ckpt").write_bytes(b"tf")                       # a TF checkpoint is named in the message
    with pytest.raises(FileNotFoundError, match="TensorFlow checkpoint"):
        vgg.load_vgg(str(tmp_path / "vgg_19.ckpt"), "cpu")


def test_synthetic_weights_are_seeded_and_he_scaled():
    a, b = vgg.synthetic_weights(123, upto="conv2_1"), vgg.synthetic_weights(123, upto="conv2_1")
    assert list(a) == ["conv1_1", "conv1_2", "conv2_1"]
    for k in a:
        assert np.array_equal(a[k][0], b[k][0])
    w = a["conv1_2"][0]
    assert abs(float(w.std()) - np.sqrt(2.0 / (9 * 64))) < 2e-3
