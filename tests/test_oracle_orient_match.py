"""Known-answer tests: IC orientation, rotated BRIEF bit order, Hamming, getBestMatch scan semantics (Q6),
stereo matching on a shifted copy, deterministic math vs libm."""
import math

import numpy as np
import pytest

from orb_slam2_ros2_amd import synth


def _pattern():
    import os
    rows = []
    inc = os.path.join(os.path.dirname(__file__), "..", "orb_slam2_ros2_amd", "csrc", "brief_pattern.inc")
    for ln in open(inc):
        if ln.lstrip().startswith("{"):
            for tok in ln.strip().rstrip(",").split("}, {"):
                rows.append([int(v) for v in tok.strip("{} ,").split(",")])
    return np.asarray(rows, np.int32)


def test_umax_table(orc):
    ex = orc.extractor(np.zeros((80, 80), np.uint8), n_features=10, n_levels=1)
    assert ex.umax().tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert 31 + 2 * sum(2 * u + 1 for u in ex.umax()[1:]) == 749


@pytest.mark.parametrize("grad,deg", [("x+", 0.0), ("y+", 90.0), ("x-", 180.0), ("y-", -90.0)])
def test_ic_angle_on_pure_gradients_and_descriptor_from_raw_template(orc, grad, deg):
    ys, xs = np.mgrid[0:80, 0:80]
    img = {"x+": xs + 60, "y+": ys + 60, "x-": 180 - xs, "y-": 180 - ys}[grad].astype(np.uint8)
    ex = orc.extractor(img, n_features=10, n_levels=1)
    theta, desc, m10, m01 = ex.describe(0, 40, 40)
    assert math.degrees(theta) == pytest.approx(deg, abs=1e-12)
    assert (m10 == 0) == (grad[0] == "y") and (m01 == 0) == (grad[0] == "x")
    # On a linear ramp the 7x7 blur is the identity (away from the border) and rotating the template by theta
    # makes every comparison reduce to x1 < x2 of the RAW template, LSB-first, 8 pairs per byte.
    pat = _pattern()
    bits = (pat[:, 0] < pat[:, 2]).astype(np.uint8)
    exp = np.packbits(bits.reshape(32, 8), axis=1, bitorder="little").reshape(32)
    assert np.array_equal(desc, exp)


def test_hamming_vectors(orc):
    z = np.zeros(32, np.uint8)
    o = np.full(32, 255, np.uint8)
    assert orc.hamming(z, z) == 0 and orc.hamming(z, o) == 256 and orc.hamming(o, o) == 0
    for bit in (0, 7, 8, 100, 255):
        a = z.copy()
        a[bit >> 3] |= 1 << (bit & 7)
        assert orc.hamming(a, z) == 1
    r = np.random.default_rng(0)
    a, b = r.integers(0, 256, 32, dtype=np.uint8), r.integers(0, 256, 32, dtype=np.uint8)
    assert orc.hamming(a, b) == int(np.unpackbits(a ^ b).sum())


def _train_with_distances(dists):
    """query = zeros; train[i] has exactly dists[i] bits set."""
    t = np.zeros((len(dists), 32), np.uint8)
    for i, d in enumerate(dists):
        bits = np.zeros(256, np.uint8)
        bits[:d] = 1
        t[i] = np.packbits(bits, bitorder="little")
    return np.zeros(32, np.uint8), t


@pytest.mark.parametrize("dists,exp", [([50, 40, 30], (2, 30, 2**31 - 1)), ([30, 40, 50], (0, 30, 40)), ([30, 30], (0, 30, 30)),
                                       ([40, 50, 30, 35], (2, 30, 35)), ([60], (0, 60, 2**31 - 1)), ([10, 10, 5, 5, 7], (2, 5, 5))])
def test_best_match_does_not_demote_the_old_minimum(orc, dists, exp):
    q, t = _train_with_distances(dists)
    bi, bd, sd, ratio = orc.best_match(q, t, np.arange(len(dists)))
    assert (bi, bd, sd) == exp
    assert ratio == np.float32(bd) / np.float32(sd)


def test_best_match_respects_candidate_order(orc):
    q, t = _train_with_distances([30, 40, 50])
    assert orc.best_match(q, t, np.array([2, 1, 0]))[:3] == (0, 30, 2**31 - 1)  # descending scan: every step is a record
    assert orc.best_match(q, t, np.array([1, 0, 2]))[:3] == (0, 30, 50)


def test_stereo_match_recovers_a_known_disparity(orc):
    L, _ = synth.stereo_pair(3, 700, 300, n_rect=150)
    d = 23
    R = np.empty_like(L)
    R[:, :-d] = L[:, d:]
    R[:, -d:] = L[:, -1:]
    el, er = orc.extractor(L, 1000), orc.extractor(R, 1000)
    lk, ld = el.extract()
    rk, rd = er.extract()
    fx, bf = 500.0, 250.0
    m, ru, dp, br, bd = el.stereo_match(er, lk, ld, rk, rd, fx, bf)
    ok = ru >= 0
    assert m == ok.sum() and m > 200
    disp = lk["x"][ok] - ru[ok]
    good = np.abs(disp - d) < 2.0 * 1.2 ** lk["octave"][ok] + 1e-3
    assert good.mean() > 0.95                      # exact copy: nearly every match sits at the true disparity
    assert np.allclose(dp[ok], (np.float32(bf) / (lk["x"][ok] - ru[ok].astype(np.float32))).astype(np.float64))
    assert np.all(dp[~ok] == -1) and np.all(ru[~ok] == -1)
    assert np.all(bd[ok] <= 75)


def test_det_math_within_one_ulp_of_libm(orc):
    r = np.random.default_rng(1)
    worst = 0
    for _ in range(20000):
        m01, m10 = int(r.integers(-2_000_000, 2_000_000)), int(r.integers(-2_000_000, 2_000_000))
        a, b = orc.atan2(m01, m10, 1), orc.atan2(m01, m10, 0)
        worst = max(worst, abs(np.float64(a).view(np.int64) - np.float64(b).view(np.int64)) if a * b > 0 else 0)
        s1, c1 = orc.sincos(b, 1)
        s0, c0 = orc.sincos(b, 0)
        assert abs(s1 - s0) <= 2.3e-16 and abs(c1 - c0) <= 2.3e-16
    assert worst <= 1
    for y, x in ((0, 5), (0, -5), (5, 0), (-5, 0), (0, 0)):
        assert orc.atan2(y, x, 1) == math.atan2(y, x)


def test_det_math_and_libm_give_identical_features(orc, kitti_pair):
    L, _ = kitti_pair
    k0, d0 = orc.extractor(L, math_mode=0).extract()
    k1, d1 = orc.extractor(L, math_mode=1).extract()
    assert np.array_equal(k0, k1) and np.array_equal(d0, d1)
