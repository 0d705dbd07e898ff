"""Known-answer tests for the oracle's restatement of the g2o SE3 projection edges:
analytic Jacobians vs central differences, Huber at the threshold, the mono/stereo information quirk."""
import numpy as np
import pytest

from orb_slam2_ros2_amd import ba_synth


@pytest.fixture(scope="module")
def prob():
    return ba_synth.make_problem(n_kf=12, n_pt=120)


def _eval(orc, p, poses=None, points=None):
    return orc.ba_eval_edges(p["poses"] if poses is None else poses, p["points"] if points is None else points, p["edge_pose"],
                             p["edge_point"], p["meas"], p["is_stereo"], p["info"], p["huber_delta"], p["fx"], p["fy"], p["cx"],
                             p["cy"], p["bf"])


def test_problem_shape(prob):
    E = prob["edge_pose"].size
    assert 300 < E < 1000 and 0.7 < prob["is_stereo"].mean() < 0.9


def test_error_is_measurement_minus_projection(orc, prob):
    out = _eval(orc, prob)
    e = 17
    T, X = prob["poses"][prob["edge_pose"][e]], prob["points"][prob["edge_point"][e]]
    q, t = T[:4], T[4:]
    R = np.array([[1 - 2 * (q[1]**2 + q[2]**2), 2 * (q[0] * q[1] - q[2] * q[3]), 2 * (q[0] * q[2] + q[1] * q[3])],
                  [2 * (q[0] * q[1] + q[2] * q[3]), 1 - 2 * (q[0]**2 + q[2]**2), 2 * (q[1] * q[2] - q[0] * q[3])],
                  [2 * (q[0] * q[2] - q[1] * q[3]), 2 * (q[1] * q[2] + q[0] * q[3]), 1 - 2 * (q[0]**2 + q[1]**2)]])
    P = R @ X + t
    u, v = prob["fx"] * P[0] / P[2] + prob["cx"], prob["fy"] * P[1] / P[2] + prob["cy"]
    exp = [prob["meas"][e, 0] - u, prob["meas"][e, 1] - v, prob["meas"][e, 2] - (u - prob["bf"] / P[2]) if prob["is_stereo"][e] else 0.0]
    assert np.allclose(out["error"][e], exp, rtol=0, atol=1e-9)
    w = prob["info"][e]
    assert out["chi2"][e] == pytest.approx(w * np.dot(out["error"][e], out["error"][e]), rel=1e-12)


def test_jacobians_match_central_differences(orc, prob):
    out = _eval(orc, prob)
    h = 1e-6
    E = prob["edge_pose"].size
    for e in range(0, E, 37):
        kp, pp = prob["edge_pose"][e], prob["edge_point"][e]
        rows = 3 if prob["is_stereo"][e] else 2
        for a in range(3):  # d e / d point
            pts_p, pts_m = prob["points"].copy(), prob["points"].copy()
            pts_p[pp, a] += h
            pts_m[pp, a] -= h
            num = (_eval(orc, prob, points=pts_p)["error"][e] - _eval(orc, prob, points=pts_m)["error"][e]) / (2 * h)
            assert np.allclose(out["j_point"][e][:rows, a], num[:rows], rtol=1e-5, atol=1e-5)
        for a in range(6):  # d e / d (omega, upsilon) via the g2o oplus: exp(delta) * T
            upd = np.zeros(6)
            upd[a] = h
            pose_p, pose_m = prob["poses"].copy(), prob["poses"].copy()
            pose_p[kp] = orc.se3_oplus(prob["poses"][kp], upd)
            pose_m[kp] = orc.se3_oplus(prob["poses"][kp], -upd)
            num = (_eval(orc, prob, poses=pose_p)["error"][e] - _eval(orc, prob, poses=pose_m)["error"][e]) / (2 * h)
            assert np.allclose(out["j_pose"][e][:rows, a], num[:rows], rtol=1e-5, atol=1e-4), (e, a)


def test_huber_weights_at_the_threshold(orc, prob):
    out = _eval(orc, prob)
    d2 = prob["huber_delta"] ** 2
    inl = out["chi2"] <= d2
    assert inl.any() and (~inl).any()
    assert np.all(out["rho"][inl, 0] == out["chi2"][inl]) and np.all(out["rho"][inl, 1] == 1.0)
    s = np.sqrt(out["chi2"][~inl])
    assert np.allclose(out["rho"][~inl, 1], prob["huber_delta"][~inl] / s, rtol=1e-15)
    assert np.allclose(out["rho"][~inl, 0], 2 * s * prob["huber_delta"][~inl] - d2[~inl], rtol=1e-15)
    # no kernel (delta <= 0): rho = (chi2, 1)
    p2 = dict(prob, huber_delta=np.zeros_like(prob["huber_delta"]))
    o2 = _eval(orc, p2)
    assert np.array_equal(o2["rho"][:, 0], o2["chi2"]) and np.all(o2["rho"][:, 1] == 1.0)


def test_mono_information_is_not_squared(prob):
    # quirk Q9 (Optimizer.cc:319 vs :301): the generator mirrors it, so mono infos are 1/sigma, stereo 1/sigma^2
    sig = np.float32(1.2) ** np.arange(8, dtype=np.float32)
    mono = prob["info"][prob["is_stereo"] == 0]
    st = prob["info"][prob["is_stereo"] == 1]
    assert set(np.float32(mono)) <= set(np.float32(1) / sig)
    assert set(np.float32(st)) <= set((np.float32(1) / sig) ** 2)


def test_depth_positive_flag(orc, prob):
    out = _eval(orc, prob)
    assert out["depth_positive"].all()
    pts = prob["points"].copy()
    pts[:, 2] -= 100.0
    assert not _eval(orc, prob, points=pts)["depth_positive"].any()


def test_normal_equation_blocks_are_consistent_with_the_edge_outputs(orc, prob):
    """H/b of constructQuadraticForm restated two ways: the oracle's accumulation vs numpy einsum over its edge outputs."""
    nk = prob["poses"].shape[0]
    fixed = np.zeros(nk, np.uint8)
    fixed[0] = 1
    fixed[nk // 2:] = 1
    s = orc.ba_build_system(**prob, pose_fixed=fixed)
    o = _eval(orc, prob)
    w = o["rho"][:, 1] * prob["info"]
    E = w.size
    Hll = np.zeros_like(s["Hll"])
    bl = np.zeros_like(s["bl"])
    Hpp = np.zeros_like(s["Hpp"])
    bp = np.zeros_like(s["bp"])
    for e in range(E):
        A, B, er = o["j_point"][e], o["j_pose"][e], o["error"][e]
        p, k = prob["edge_point"][e], prob["edge_pose"][e]
        Hll[p] += w[e] * A.T @ A
        bl[p] -= w[e] * A.T @ er
        if not fixed[k]:
            Hpp[k] += w[e] * B.T @ B
            bp[k] -= w[e] * B.T @ er
            assert np.allclose(s["Hpl"][e], w[e] * B.T @ A, rtol=1e-12, atol=1e-9)
        else:
            assert not s["Hpl"][e].any()
    for a, b in ((s["Hll"], Hll), (s["bl"], bl), (s["Hpp"], Hpp), (s["bp"], bp)):
        assert np.allclose(a, b, rtol=1e-11, atol=1e-7)
    assert not s["Hpp"][0].any() and s["Hpp"][1].any()
    assert s["chi2_robust"] == pytest.approx(o["rho"][:, 0].sum(), rel=1e-13)
    # Schur-reducible: every H_ll block of an observed point is symmetric positive definite
    obs = np.unique(prob["edge_point"])
    assert np.all(np.linalg.eigvalsh(s["Hll"][obs]) > 0)
