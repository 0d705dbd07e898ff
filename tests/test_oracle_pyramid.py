"""Known-answer tests that pin the oracle's restatement of cv::resize / cv::GaussianBlur
 against an independent numpy formulation and hand-computable cases."""
import numpy as np
import pytest


def np_resize_linear(src, dw, dh):
    """Independent restatement of OpenCV's 8-bit INTER_LINEAR (vectorised numpy, int64 arithmetic)."""
    sh, sw = src.shape

    def axis(s, d):
        scale = 1.0 / (np.float64(d) / s)
        i = np.arange(d, dtype=np.float64)
        f = ((i + 0.5) * scale - 0.5).astype(np.float32)
        o = np.floor(f).astype(np.int64)
        f = (f - o.astype(np.float32)).astype(np.float32)
        return o, f

    xo, fx = axis(sw, dw)
    fx = np.where((xo < 0) | (xo >= sw - 1), np.float32(0), fx)
    xo = np.clip(xo, 0, sw - 1)
    a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64)
    a1 = np.rint(fx * np.float32(2048)).astype(np.int64)
    yo, fy = axis(sh, dh)
    b0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int64)
    b1 = np.rint(fy * np.float32(2048)).astype(np.int64)
    y0 = np.clip(yo, 0, sh - 1)
    y1 = np.clip(yo + 1, 0, sh - 1)
    x1 = np.clip(xo + 1, 0, sw - 1)
    S = src.astype(np.int64)
    H0 = S[y0][:, xo] * a0 + S[y0][:, x1] * a1
    H1 = S[y1][:, xo] * a0 + S[y1][:, x1] * a1
    v = (((b0[:, None] * (H0 >> 4)) >> 16) + ((b1[:, None] * (H1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def np_gauss7(src, taps):
    t = np.asarray(taps, np.int64)
    h, w = src.shape
    p = np.pad(src.astype(np.int64), 3, mode="reflect")  # numpy 'reflect' == BORDER_REFLECT_101
    row = sum(t[k] * p[3:3 + h, k:k + w] for k in range(7))
    row = np.minimum(row, 65535)
    rp = np.pad(row, ((3, 3), (0, 0)), mode="reflect")
    col = sum(t[k] * rp[k:k + h, :] for k in range(7))
    return np.minimum((col + 0x8000) >> 16, 255).astype(np.uint8)


TAPS_A = [18, 34, 48, 56, 48, 34, 18]
TAPS_B = [18, 34, 49, 55, 49, 34, 18]


@pytest.mark.parametrize("shape,dst", [((376, 1241), (1034, 313)), ((376, 1241), (346, 105)), ((480, 640), (533, 400)),
                                       ((61, 97), (53, 40)), ((40, 40), (40, 40))])
def test_resize_matches_numpy_restatement(orc, rng, shape, dst):
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    out = orc.resize(img, dst[0], dst[1])
    assert np.array_equal(out, np_resize_linear(img, dst[0], dst[1]))


def test_resize_identity_and_constants(orc, rng):
    img = rng.integers(0, 256, (50, 70), dtype=np.uint8)
    assert np.array_equal(orc.resize(img, 70, 50), img)  # scale 1: taps (2048, 0)
    for v in (0, 1, 127, 254, 255):
        flat = np.full((60, 90), v, np.uint8)
        assert np.all(orc.resize(flat, 75, 50) == v)


def test_resize_hand_computed_pixel(orc):
    # 4x1 -> 2x1 (scale 2): dx=0 -> fx=0.5, sx=0 -> (S0*1024+S1*1024); vertical b=(2048,0) with the row clipped
    img = np.array([[10, 30, 200, 100]], np.uint8)
    out = orc.resize(img, 2, 1)
    h0 = 10 * 1024 + 30 * 1024
    h1 = 200 * 1024 + 100 * 1024
    exp = [(((2048 * (h >> 4)) >> 16) + 0 + 2) >> 2 for h in (h0, h1)]
    assert out.tolist() == [exp] == [[20, 150]]


@pytest.mark.parametrize("variant,taps", [(0, TAPS_A), (1, TAPS_B)])
def test_blur_matches_numpy_restatement(orc, rng, variant, taps):
    for shape in ((105, 346), (38, 38), (64, 200)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        assert np.array_equal(orc.gauss7(img, variant), np_gauss7(img, taps))


def test_blur_taps_and_impulse(orc):
    assert sum(TAPS_A) == 256 and sum(TAPS_B) == 257
    img = np.zeros((41, 41), np.uint8)
    img[20, 20] = 255
    out = orc.gauss7(img, 0)
    t = np.asarray(TAPS_A, np.int64)
    exp = ((np.outer(t, t) * 255 + 0x8000) >> 16).astype(np.uint8)
    assert np.array_equal(out[17:24, 17:24], exp)
    assert out.sum() == exp.sum()  # nothing leaks outside the 7x7 support


def test_blur_flat_and_ramp_are_fixed_points(orc):
    for v in (0, 7, 128, 255):
        flat = np.full((50, 60), v, np.uint8)
        assert np.all(orc.gauss7(flat, 0) == v)
    assert np.all(orc.gauss7(np.full((50, 60), 255, np.uint8), 1) == 255)  # variant B saturates, still 255
    ramp = np.tile(np.arange(100, dtype=np.uint8), (50, 1))
    out = orc.gauss7(ramp, 0)
    assert np.array_equal(out[:, 3:-3], ramp[:, 3:-3])  # symmetric taps reproduce a linear ramp away from the border


def test_blur_reflect101_border(orc):
    # column 0 of a horizontal step: taps see gfedcb|abcdefg -> pixels (3,2,1,0,1,2,3)
    row = np.array([0, 0, 0, 100, 100, 100, 100, 100, 100, 100] + [100] * 40, np.uint8)
    img = np.tile(row, (45, 1))
    out = orc.gauss7(img, 0)
    acc0 = 18 * 100 + 34 * 0 + 48 * 0 + 56 * 0 + 48 * 0 + 34 * 0 + 18 * 100
    assert out[20, 0] == (acc0 * 256 + 0x8000) >> 16
