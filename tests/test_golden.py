"""The oracle against the committed golden vectors (tests/golden/golden_v1.json, made by tools/make_golden.py).
These pin the oracle + synthetic generator; the GPU parity tests compare the HIP path with the same vectors."""
import hashlib
import json
import os

import numpy as np
import pytest

from orb_slam2_ros2_amd import ba_synth, synth

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.json")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("f", [0, 1, 7])
def test_kitti_frames(orc, f):
    g = G["frames"][f"kitti_{f}"]
    L, R = synth.stereo_pair(f)
    assert sha(L) == g["left_sha"] and sha(R) == g["right_sha"]
    r = orc.stereo_frame(L, R, fx=718.856, bf=718.856 * 0.537166, math_mode=0, threads=2)
    assert (len(r["lk"]), len(r["rk"]), r["n_matches"]) == (g["n_left"], g["n_right"], g["n_matches"])
    assert sha(r["lk"]) == g["lk_sha"] and sha(r["ld"]) == g["ld_sha"]
    assert sha(r["rk"]) == g["rk_sha"] and sha(r["rd"]) == g["rd_sha"]
    assert sha(r["right_u"]) == g["right_u_sha"] and sha(r["depth"]) == g["depth_sha"]
    assert r["ld"][:2].tolist() == g["first_desc"]
    # deterministic-math mode gives the same bytes
    r1 = orc.stereo_frame(L, R, fx=718.856, bf=718.856 * 0.537166, math_mode=1, threads=1)
    assert sha(r1["lk"]) == g["lk_sha"] and sha(r1["ld"]) == g["ld_sha"] and sha(r1["right_u"]) == g["right_u_sha"]


def test_level_geometry_matches_survey_table(orc):
    ex = orc.extractor(np.zeros((376, 1241), np.uint8))
    dims = [ex.level_info(l)[:2] for l in range(8)]
    assert dims == [(1241, 376), (1034, 313), (862, 261), (718, 218), (598, 181), (499, 151), (416, 126), (346, 105)]
    assert [ex.level_info(l)[3] for l in range(8)] == [434, 362, 302, 252, 210, 175, 146, 119]
    ex = orc.extractor(np.zeros((480, 640), np.uint8), n_features=1000)
    assert [list(ex.level_info(l)[:2]) for l in range(8)] == G["frames"]["tum_0"]["level_dims"]
    assert [ex.level_info(l)[3] for l in range(8)] == G["frames"]["tum_0"]["quotas"] == [217, 181, 151, 126, 105, 88, 73, 59]


def test_tum_frame_and_sparse_frame(orc):
    g = G["frames"]["tum_0"]
    img = synth.mono_image(0)
    assert sha(img) == g["img_sha"]
    k, d = orc.extractor(img, n_features=1000).extract()
    assert (len(k), sha(k), sha(d)) == (g["n"], g["k_sha"], g["d_sha"])
    gs = G["frames"]["sparse_5"]
    Ls, _ = synth.stereo_pair(5, sparse=True)
    ks, _ = orc.extractor(Ls).extract()
    assert sha(Ls) == gs["img_sha"] and len(ks) == gs["n"] == 0  # quirk Q3: fewer candidates than quota => nothing


def test_image_size_error(orc):
    with pytest.raises(ValueError):
        orc.extractor(np.zeros((100, 130), np.uint8))  # level 7 would be 36x28 < 38 px (ORBExtractor.cc:310-314)


def test_cfg3_bruteforce(orc):
    g = G["cfg3"]
    q, t = synth.descriptors_cfg3()
    assert sha(q) == g["q_sha"] and sha(t) == g["t_sha"]
    bi, bd, sd = orc.match_bruteforce(q, t)
    assert (sha(bi), sha(bd), sha(sd)) == (g["best_idx_sha"], g["best_dist_sha"], g["second_sha"])


def test_cfg5_ba(orc):
    g = G["cfg5_ba"]
    p = ba_synth.make_problem()
    assert p["edge_pose"].size == g["n_edges"] and sha(p["meas"]) == g["meas_sha"]
    o = orc.ba_eval_edges(p["poses"], p["points"], p["edge_pose"], p["edge_point"], p["meas"], p["is_stereo"], p["info"],
                          p["huber_delta"], p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
    assert o["chi2"].sum() == pytest.approx(g["chi2_sum"], rel=1e-12)
    assert np.abs(o["j_pose"]).sum() == pytest.approx(g["jpose_abs_sum"], rel=1e-12)


# ---- golden_v2: the widened rows (local BA, pose-only, frame glue) ----------------------------------------------------------
GOLDEN_DIR = os.path.join(os.path.dirname(__file__), "golden")
G2 = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_v2.json")))
TUM = dict(fx=520.908620, fy=521.007327, cx=325.141442, cy=249.701764, k1=0.231222, k2=-0.784899, p1=-0.003257, p2=-0.000105, k3=0.917205, bf=40.0)


def _lba_problem():
    pr = ba_synth.make_problem(seed=3, n_kf=12, n_pt=400, with_truth=True)
    fixed = np.zeros(12, np.uint8)
    fixed[:2] = 1
    pr["poses"][:2] = pr["poses_true"][:2]
    return pr, fixed


def _color_image(seed, w=640, h=480):
    g = synth.mono_image(seed, w, h)
    rng = np.random.default_rng(seed)
    img = np.stack([g, np.roll(g, 3, 1), (255 - g)], 2).astype(np.int32) + rng.integers(-6, 7, (h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def _check_lba(r):
    g = G2["lba"]
    assert r["iters"].tolist() == g["iters"] and int(r["level"].sum()) == g["n_level1"] and int(r["bad"].sum()) == g["n_bad"]
    assert np.allclose(r["poses"].ravel(), g["poses"], rtol=0, atol=1e-8) and np.allclose(r["points"][:20].ravel(), g["points_head"], rtol=0, atol=1e-8)
    assert abs(r["chi2"].sum() - g["chi2_sum"]) < 1e-6 * g["chi2_sum"]


def test_v2_oracle_local_ba_pose_only_and_glue(orc):
    pr, fixed = _lba_problem()
    r = orc.ba_local_optimize(pr, fixed)
    _check_lba(r)
    assert sha(r["level"]) == G2["lba"]["level_sha"] and sha(r["bad"]) == G2["lba"]["bad_sha"]
    p = ba_synth.make_pose_problem()
    n_good, pose, inl = orc.pose_only_optimize(p["Xw"], p["meas"], p["info"], p["sigma2"], p["pose"], p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
    assert n_good == G2["pose_only"]["n_good"] and np.allclose(pose, G2["pose_only"]["pose"], rtol=0, atol=1e-9)
    assert sha(inl.astype(np.uint8)) == G2["pose_only"]["inlier_sha"]
    img = _color_image(3)
    assert sha(img) == G2["glue"]["img_sha"]
    assert sha(orc.cvt_gray(img, 1)) == G2["glue"]["gray_rgb_sha"] and sha(orc.cvt_gray(img, 2)) == G2["glue"]["gray_bgr_sha"]
    K = np.array([TUM[q] for q in ("fx", "fy", "cx", "cy")], np.float32)
    D = np.array([TUM[q] for q in ("k1", "k2", "p1", "p2", "k3")], np.float32)
    pts = np.stack([np.linspace(30, 610, 40), np.linspace(25, 455, 40)[::-1]], 1).astype(np.float32)
    assert sha(orc.undistort_points(pts, K, D)) == G2["glue"]["undistort_sha"]


@pytest.mark.gpu
def test_v2_device_against_golden():
    from orb_slam2_ros2_amd._lib import Context
    ctx = Context(640, 480, n_features=1000, max_images=1)
    pr, fixed = _lba_problem()
    _check_lba(ctx.ba_local_optimize(pr, fixed))
    p = ba_synth.make_pose_problem()
    n_good, pose, inl = ctx.pose_only_optimize(p["Xw"], p["meas"], p["info"], p["sigma2"], p["pose"], p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
    assert abs(n_good - G2["pose_only"]["n_good"]) <= 1 and np.allclose(pose, G2["pose_only"]["pose"], rtol=0, atol=1e-6)
    img = _color_image(3)
    ctx.extract_color(img, 1)
    assert sha(ctx.pyramid(0, 0, False)) == G2["glue"]["gray_rgb_sha"]
    ctx.extract_color(img, 2)
    assert sha(ctx.pyramid(0, 0, False)) == G2["glue"]["gray_bgr_sha"]
    ctx.close()


def test_golden_map_pb_written_by_libprotobuf():
    """tests/golden/map_small.pb was serialised by the real protobuf runtime (tools/make_golden_map.py): host/map_pb.hpp must read
    it, write it back byte for byte, and build the local-map graph recorded next to it.  No protobuf import here."""
    import hashlib

    from orb_slam2_ros2_amd import _lib
    pb = open(os.path.join(GOLDEN_DIR, "map_small.pb"), "rb").read()
    meta = json.load(open(os.path.join(GOLDEN_DIR, "map_small.json")))
    assert hashlib.sha256(pb).hexdigest() == meta["sha256"]
    assert _lib.map_pb_summary(pb) == meta["summary"]
    assert _lib.map_pb_reencode(pb) == pb
    g = _lib.map_local_graph(pb, meta["graph_kf_id"])
    for key, want in meta["graph"].items():
        if key == "n_group":
            assert g[key] == want
        elif key in ("poses",):
            assert np.abs(np.asarray(g[key]) - np.asarray(want)).max() < 1e-15
        else:
            assert (np.asarray(g[key]) == np.asarray(want, dtype=np.asarray(g[key]).dtype)).all(), key


# ---- golden_v3: the BASELINE config-5 problem the bench's `ba` leg times -----------------------------------------------------------
G3 = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_v3.json")))


def cfg5_local_ba_problem():
    pr = ba_synth.make_problem(seed=42, n_kf=60, n_pt=3000, with_truth=True)
    fixed = np.zeros(60, np.uint8)
    fixed[:20] = 1
    pr["poses"][:20] = pr["poses_true"][:20]
    return pr, fixed


def check_cfg5_lba(r, atol=1e-8):
    g = G3["cfg5_lba"]
    assert r["iters"].tolist() == g["iters"] and abs(int(r["level"].sum()) - g["n_level1"]) <= 1 and abs(int(r["bad"].sum()) - g["n_bad"]) <= 1
    assert np.allclose(r["poses"].ravel(), g["poses"], rtol=0, atol=atol) and np.allclose(r["points"][:20].ravel(), g["points_head"], rtol=0, atol=atol)
    assert abs(r["chi2"].sum() - g["chi2_sum"]) < 1e-6 * g["chi2_sum"]


def check_cfg5_system(s):
    for k, want in G3["cfg5_system"].items():
        assert np.abs(np.asarray(s[k], np.float64)).sum() == pytest.approx(want, rel=1e-9), k


def _cfg5_system_args():
    p = ba_synth.make_problem()
    fx = np.zeros(p["poses"].shape[0], np.uint8)
    fx[0] = 1
    fx[30:] = 1
    return p, fx


def test_v3_oracle_cfg5_local_ba_and_system(orc):
    pr, fixed = cfg5_local_ba_problem()
    assert pr["edge_pose"].size == G3["cfg5_lba"]["n_edges"]
    check_cfg5_lba(orc.ba_local_optimize(pr, fixed))
    p, fx = _cfg5_system_args()
    check_cfg5_system(orc.ba_build_system(**p, pose_fixed=fx))


@pytest.mark.gpu
def test_v3_device_cfg5_local_ba_and_system():
    from orb_slam2_ros2_amd._lib import Context
    ctx = Context(640, 480, n_features=1000, max_images=1)
    pr, fixed = cfg5_local_ba_problem()
    check_cfg5_lba(ctx.ba_local_optimize(pr, fixed), atol=1e-7)
    p, fx = _cfg5_system_args()
    check_cfg5_system(ctx.ba_build_system(**p, pose_fixed=fx))
    ctx.close()


def test_golden_v5_sequence_digests(orc):
    """tests/golden/golden_v5.json (tools/make_golden_v5.py): the oracle's pair digest of EVERY frame of the 4541-pair sequence -- what bench.py
    verifies its 512 distinct pairs per step against.  Here: the fixture's shape, its agreement with golden_v1 on frames 0 .. 127, and the
    oracle itself on a handful of frames across the range."""
    from orb_slam2_ros2_amd.digest import pair_digest
    g5 = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_v5.json")))
    n, hx = g5["n_frames"], g5["hex_chars"]
    assert n == 4541 == len(g5["pairs"]) and hx == 24 and all(len(d) == hx for d in g5["pairs"])
    assert len(set(g5["pairs"])) == n                        # every frame is a different image pair
    assert all(G["bench_pairs"][str(f)][:hx] == g5["pairs"][f] for f in range(128))
    for f in (128, 511, 2222, 4540):
        L, R = synth.stereo_pair(f)
        r = orc.stereo_frame(L, R, fx=718.856, bf=718.856 * 0.537166, math_mode=0, threads=2)
        assert pair_digest(r["lk"], r["ld"], r["rk"], r["rd"], r["right_u"], r["depth"], r["n_matches"])[:hx] == g5["pairs"][f], f"frame {f}"
