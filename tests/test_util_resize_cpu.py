"""util.resize (reference util.py:187-207 -> skimage.transform.resize(order, mode='constant', anti_aliasing=True)): scikit-image
is not installed here, so the restatement is held to the properties its published algorithm has."""
import numpy as np

from neural_flow_style_amd import util


def test_equal_size_is_an_exact_copy_and_the_range_is_kept():
    rng = np.random.RandomState(0)
    img = rng.uniform(0, 255, (17, 23, 3)).astype(np.float32)       # 0..255: normalised to [0,1] inside and back
    for order in (0, 1, 3):
        out = util.resize(img, [17, 23], order=order)
        np.testing.assert_allclose(out, img, rtol=0, atol=2e-5 * 255)
    up = util.resize(img, [40, 50], order=3)
    assert up.shape == (40, 50, 3) and up.min() >= img.min() - 1e-3 and up.max() <= img.max() + 1e-3   # clip=True


def test_interior_partition_of_unity_and_linear_precision():
    # cubic convolution (Catmull-Rom) reproduces constants and linear ramps; the Gaussian anti-aliasing filter too, away
    # from the zero-padded border (mode='constant')
    H, W = 64, 96
    const = np.full((H, W), 0.7, np.float32)
    const[40:, 60:] = 0.1                       # (two levels: a constant image would be clipped back to itself)
    out = util.resize(const, [32, 48], order=3)
    np.testing.assert_allclose(out[4:16, 4:24], 0.7, atol=1e-6)
    np.testing.assert_allclose(out[24:-4, 34:-4], 0.1, atol=1e-6)
    # (at the border the zero padding enters both the Gaussian and the cubic taps -- with the negative outer lobe of the
    # Catmull-Rom kernel the corner comes out at 0.703 before the clip to the input's range)
    assert abs(out[0, 0] - 0.7) < 1e-6
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    ramp = (0.2 + 0.004 * yy + 0.003 * xx).astype(np.float32)
    out = util.resize(ramp, [32, 48], order=3)
    # output pixel (i, j) sits at input coordinates ((i + .5) * 2 - .5, (j + .5) * 2 - .5)
    ii, jj = np.meshgrid(np.arange(32), np.arange(48), indexing="ij")
    want = 0.2 + 0.004 * ((ii + 0.5) * 2 - 0.5) + 0.003 * ((jj + 0.5) * 2 - 0.5)
    np.testing.assert_allclose(out[4:-4, 4:-4], want[4:-4, 4:-4], atol=2e-6)
    # upscaling a ramp by 1.5: no pre-filter, pure cubic convolution
    out = util.resize(ramp, [96, 144], order=3)
    ii, jj = np.meshgrid(np.arange(96), np.arange(144), indexing="ij")
    want = 0.2 + 0.004 * ((ii + 0.5) / 1.5 - 0.5) + 0.003 * ((jj + 0.5) / 1.5 - 0.5)
    np.testing.assert_allclose(out[4:-4, 4:-4], want[4:-4, 4:-4], atol=2e-6)


def test_catmull_rom_taps_at_a_half_pixel():
    # a unit impulse sampled half way between pixels spreads as (-1/16, 9/16, 9/16, -1/16): the Catmull-Rom kernel,
    # not the cubic B-spline's (1/48, 23/48, 23/48, 1/48)
    M = util._interp_matrix(8, 16, 3)            # scale 0.5: output 2k+... lands on x.25 / x.75; use a direct half-pixel case
    M2 = util._interp_matrix(9, 8, 3)            # src = (i + .5) * 9/8 - .5: i = 3 -> 3.4375 (not needed further)
    assert abs(M.sum(1)[4:-4] - 1.0).max() < 1e-12 and abs(M2.sum(1)[2:-2] - 1.0).max() < 1e-12
    x = 0.5
    w = [-0.5 * x ** 3 + x ** 2 - 0.5 * x, 1.5 * x ** 3 - 2.5 * x ** 2 + 1, -1.5 * x ** 3 + 2 * x ** 2 + 0.5 * x,
         0.5 * x ** 3 - 0.5 * x ** 2]
    np.testing.assert_allclose(w, [-1 / 16, 9 / 16, 9 / 16, -1 / 16])
    Mh = util._interp_matrix(10, 10, 3)
    np.testing.assert_allclose(Mh, np.eye(10), atol=1e-15)
