"""bench.py's extra legs beside the stereo step: config 3 (Hamming brute force), the BA half (config 5) and the one-frame latency leg."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

from .common import *  # noqa: F401,F403
from .common import _cpu_ms, _kernel_us, _oracle_fast, _sha, _stats_ms  # noqa: F401


def cfg3_leg(device_id):
    """BASELINE config 3: 2000 x 2000 Hamming-256 brute force (ORBMatcher::getBestMatch over all train descriptors,
    src/ORBMatcher.cc:941-990), results bit-exact against tests/golden/golden_v1.json before any number is reported."""
    from orb_slam2_ros2_amd import synth
    from orb_slam2_ros2_amd._lib import Context
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_v1.json")))["cfg3"]
    ctx = Context(W, H, NFEAT, NLEVELS, SCALE, TH_HI, TH_LO, device_id=device_id, max_images=2)
    q, t = synth.descriptors_cfg3()
    bi, bd, sd = ctx.match_bruteforce(q, t)
    if (_sha(bi), _sha(bd), _sha(sd)) != (gold["best_idx_sha"], gold["best_dist_sha"], gold["second_sha"]):
        raise SystemExit("bench.py: cfg3 leg: results differ from the golden vectors")
    us = _kernel_us(ctx, "match", lambda: ctx.match_bruteforce(q, t))
    host = _stats_ms(lambda: ctx.match_bruteforce(q, t), 50)
    orc = _oracle_fast()
    cand = np.arange(t.shape[0], dtype=np.uint32)
    t0 = time.perf_counter()
    for i in range(100):
        orc.best_match(q[i], t, cand)
    cpu_ms = (time.perf_counter() - t0) / 100 * q.shape[0] * 1e3
    ctx.close()
    alg = 2 * 2000 * 32 + 2000 * 12
    pairs = q.shape[0] * t.shape[0]
    return {"workload": "2000 x 2000 Hamming-256 brute force, best / second best per query (quirk Q6), bit-exact vs golden_v1",
            "verified": True, "kernel_us": us, "Gpair_per_s": pairs / (us * 1e-6) / 1e9 if us else None,
            "algorithmic_bytes": alg, "GBps": alg / (us * 1e-6) / 1e9 if us else None,
            "bound": "neither HBM (152 KB) nor issue: one launch of 500 workgroups, latency of a single wave pass",
            "host_call": dict(host, what="host descriptors in, host results out (PCIe both ways included)"),
            "cpu_baseline": {"ms": cpu_ms, "cores": 1, "kind": "port"}}


def ba_leg(device_id):
    """The BA half of north_star on the BASELINE config-5 problem (60 keyframes, 3000 points, 15 597 edges): g2o edge
    evaluation (src/Optimizer.cc:296-330 set-up, computeError / linearizeOplus / Huber), the normal-equation build, the local BA
    (Optimizer.cc:336-361: optimize(5) + optimize(10)) and OptimizePoseOnly (:33-178) -- each checked against the golden vectors first."""
    from orb_slam2_ros2_amd import ba_synth
    from orb_slam2_ros2_amd._lib import Context
    g1 = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_v1.json")))["cfg5_ba"]
    g2 = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_v2.json")))["pose_only"]
    g3 = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_v3.json")))
    ctx = Context(640, 480, n_features=1000, device_id=device_id, max_images=1)
    orc = _oracle_fast()
    out = {"workload": "BASELINE config 5: synthetic local map, 60 keyframes / 3000 points / 15 597 edges (80 % stereo), TUM intrinsics"}
    # edge evaluation
    p = ba_synth.make_problem()
    E = int(p["edge_pose"].size)
    r = ctx.ba_eval_edges(**p)
    if E != g1["n_edges"] or abs(r["chi2"].sum() - g1["chi2_sum"]) > 1e-12 * g1["chi2_sum"] or \
            abs(np.abs(r["j_pose"]).sum() - g1["jpose_abs_sum"]) > 1e-10 * g1["jpose_abs_sum"]:
        raise SystemExit("bench.py: ba leg: edge evaluation differs from the golden vectors")
    alg = 304 * E + p["poses"].shape[0] * 56 + p["points"].shape[0] * 24
    us = _kernel_us(ctx, "ba", lambda: ctx.ba_eval_edges(**p))
    out["edge_eval"] = {"edges": E, "kernel_us": us, "algorithmic_bytes": alg, "GBps": alg / (us * 1e-6) / 1e9 if us else None,
                        "hbm_frac": alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS if us else None,
                        "host_call": dict(_stats_ms(lambda: ctx.ba_eval_edges(**p), 30), what="host arrays in, 4.8 MB of results out over PCIe"),
                        "cpu_baseline": {"ms": _cpu_ms(lambda: orc.ba_eval_edges(**p)), "cores": 1, "kind": "port"}, "verified": True}
    # ... and the same kernel at a size that fills the machine: the edge list tiled 64 times (the same poses and points; 61 workgroups of
    # the config-5 problem occupy a quarter of the chip for two launch floors, which says nothing about the kernel's memory behaviour)
    rep = 64
    pl = dict(p)
    for k in ("edge_pose", "edge_point", "meas", "is_stereo", "info", "huber_delta"):
        pl[k] = np.ascontiguousarray(np.concatenate([p[k]] * rep))
    rl = ctx.ba_eval_edges(**pl)
    if abs(rl["chi2"].sum() - rep * g1["chi2_sum"]) > 1e-11 * rep * g1["chi2_sum"] or not np.array_equal(rl["chi2"][:E], r["chi2"]) or \
            not np.array_equal(rl["chi2"][-E:], r["chi2"]):
        raise SystemExit("bench.py: ba leg: the tiled edge evaluation differs from the golden vectors")
    alg_l = 304 * E * rep + p["poses"].shape[0] * 56 + p["points"].shape[0] * 24
    us_l = _kernel_us(ctx, "ba", lambda: ctx.ba_eval_edges(**pl), n=3)
    out["edge_eval_tiled"] = {"edges": E * rep, "kernel_us": us_l, "algorithmic_bytes": alg_l, "GBps": alg_l / (us_l * 1e-6) / 1e9 if us_l else None,
                              "hbm_frac": alg_l / (us_l * 1e-6) / 1e9 / HBM_PEAK_GBPS if us_l else None, "verified": True,
                              "what": "the config-5 edge list 64 times over: the edge kernel with the chip full (SURVEY 8d: 304 B per edge)"}
    del rl, pl
    # normal equations
    fx = np.zeros(p["poses"].shape[0], np.uint8)
    fx[0] = 1
    fx[30:] = 1
    sysd = ctx.ba_build_system(**p, pose_fixed=fx)
    for k, want in g3["cfg5_system"].items():
        if abs(np.abs(np.asarray(sysd[k], np.float64)).sum() - want) > 1e-9 * want:
            raise SystemExit(f"bench.py: ba leg: normal-equation block {k} differs from the golden vectors")
    out["build_system"] = {"kernel_us": _kernel_us(ctx, "ba", lambda: ctx.ba_build_system(**p, pose_fixed=fx)),
                           "host_call": _stats_ms(lambda: ctx.ba_build_system(**p, pose_fixed=fx), 30),
                           "cpu_baseline": {"ms": _cpu_ms(lambda: orc.ba_build_system(**p, pose_fixed=fx)), "cores": 1, "kind": "port"},
                           "verified": True}
    # local BA
    pr = ba_synth.make_problem(seed=42, n_kf=60, n_pt=3000, with_truth=True)
    fixed = np.zeros(60, np.uint8)
    fixed[:20] = 1
    pr["poses"][:20] = pr["poses_true"][:20]
    r = ctx.ba_local_optimize(pr, fixed)
    gl = g3["cfg5_lba"]
    if r["iters"].tolist() != gl["iters"] or not np.allclose(r["poses"].ravel(), gl["poses"], rtol=0, atol=1e-7) or \
            abs(r["chi2"].sum() - gl["chi2_sum"]) > 1e-6 * gl["chi2_sum"] or abs(int(r["bad"].sum()) - gl["n_bad"]) > 1:
        raise SystemExit("bench.py: ba leg: local BA differs from the golden vectors")
    out["local_optimize"] = dict(_stats_ms(lambda: ctx.ba_local_optimize(pr, fixed), 20, warm=2),
                                 what="orbfe_ba_local_optimize: optimize(5) + re-classification + optimize(10), 40 free keyframes, host arrays "
                                      "in, host results out", iterations=gl["iters"],
                                 cpu_baseline={"ms": _cpu_ms(lambda: orc.ba_local_optimize(pr, fixed), 2.0, 3), "cores": 1, "kind": "port"},
                                 verified=True)
    # the same over the window size: past 42 free keyframes the blocked multi-workgroup Cholesky (fp64 MFMA) takes the reduced system; the
    # 43 / 64 / 100 problems are 50 points per keyframe and 10 fixed keyframes (tools/lba_sizes.py), the 300 one is the GPU suite's
    for nf_w in (43, 64, 100):
        n_kf = nf_w + 10
        w = ba_synth.make_problem(seed=100 + nf_w, n_kf=n_kf, n_pt=50 * n_kf, with_truth=True)
        fw = np.zeros(n_kf, np.uint8)
        fw[:10] = 1
        w["poses"][:10] = w["poses_true"][:10]
        out[f"local_optimize_{nf_w}_free_keyframes"] = dict(_stats_ms(lambda: ctx.ba_local_optimize(w, fw), 5, warm=1), edges=int(w["edge_pose"].size),
                                                            cpu_baseline={"ms": _cpu_ms(lambda: orc.ba_local_optimize(w, fw), 1.0, 1), "cores": 1, "kind": "port"})
    big = ba_synth.make_problem(seed=13, n_kf=310, n_pt=4000, with_truth=True)
    fb = np.zeros(310, np.uint8)
    fb[:10] = 1
    big["poses"][:10] = big["poses_true"][:10]
    gb = ctx.ba_local_optimize(big, fb)
    t0 = time.perf_counter()
    ob = orc.ba_local_optimize(big, fb)
    cpu_big_ms = (time.perf_counter() - t0) * 1e3
    if tuple(gb["iters"]) != tuple(ob["iters"]) or np.abs(gb["points"] - ob["points"]).max() > 1e-7:
        raise SystemExit("bench.py: ba leg: the 300-keyframe local BA differs from the oracle")
    out["local_optimize_300_free_keyframes"] = dict(_stats_ms(lambda: ctx.ba_local_optimize(big, fb), 5, warm=1), edges=int(big["edge_pose"].size),
                                                    cpu_baseline={"ms": cpu_big_ms, "cores": 1, "kind": "port", "sample": "one call"}, verified=True)
    # pose only
    pp = ba_synth.make_pose_problem()
    a = (pp["Xw"], pp["meas"], pp["info"], pp["sigma2"], pp["pose"], pp["fx"], pp["fy"], pp["cx"], pp["cy"], pp["bf"])
    n_good, pose, _ = ctx.pose_only_optimize(*a)
    if abs(n_good - g2["n_good"]) > 1 or not np.allclose(pose, g2["pose"], rtol=0, atol=1e-6):
        raise SystemExit("bench.py: ba leg: pose-only optimisation differs from the golden vectors")
    out["pose_only"] = dict(_stats_ms(lambda: ctx.pose_only_optimize(*a), 50), edges=int(len(pp["info"])),
                            kernel_us=_kernel_us(ctx, "ba", lambda: ctx.pose_only_optimize(*a), n=20),
                            what="orbfe_pose_only_optimize: 4 x optimize(10) on one SE3 vertex, host arrays in and out",
                            cpu_baseline={"ms": _cpu_ms(lambda: orc.pose_only_optimize(*a)), "cores": 1, "kind": "port"}, verified=True)
    ctx.close()
    return out


def cpp_latency_harness(n, L, R):
    """tests/cpp/test_dropin `latency`: the C++ drop-in's one-frame call shapes, n frames each, host cv::Mat in, host results out.  THE harness
    of every one-frame figure (bench.py's latency leg, DESIGN 4.10, tools/exp/latency_contexts.sh): built and run here as a child process.
    Returns (fields of the LATENCY_OK line, {call shape: [n, p50, p90, p99, p99.9, max, extract p50, extract p99] in us}, hw-queue setting)."""
    import subprocess
    import tempfile
    tmp = tempfile.mkdtemp(prefix="orbfe_lat_")
    exe = os.path.join(tmp, "test_dropin")
    pkg = os.path.join(ROOT, "orb_slam2_ros2_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "tests", "cpp", "stubs"), "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "test_dropin.cpp"), "-L" + pkg, "-lorbfe_hip", "-pthread",
                           "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"])
    L.tofile(os.path.join(tmp, "L.raw"))
    R.tofile(os.path.join(tmp, "R.raw"))
    # (the C++ child gets the same setting as this leg: the runtime's default unless ORBFE_LATENCY_HW_QUEUES asks for a value)
    env = dict(os.environ)
    if os.environ.get("ORBFE_LATENCY_HW_QUEUES"):
        env["GPU_MAX_HW_QUEUES"] = os.environ["ORBFE_LATENCY_HW_QUEUES"]
    r = subprocess.run([exe, "latency", os.path.join(tmp, "L.raw"), os.path.join(tmp, "R.raw"), str(W), str(H), str(n)],
                       capture_output=True, text=True, timeout=900, env=env)
    lines = r.stdout.strip().split("\n")
    f = lines[0].split() if lines else []
    if r.returncode != 0 or not f or f[0] != "LATENCY_OK":
        raise RuntimeError((r.stdout + r.stderr)[-400:])
    q = {l.split()[1]: [float(v) for v in l.split()[2:]] for l in lines[1:] if l.startswith("LATQ")}
    return f, q, env.get("GPU_MAX_HW_QUEUES", "unset (runtime default)")


def latency_leg(device_id, n=500, n_cpp=2000):
    """One stereo pair from host images to host results, in the two call shapes a caller has: (a) one batched call for both eyes +
    the match; (b) the reference's own -- Frame::Frame builds two ORBExtractor objects and runs extract() on two std::threads
    (src/Frame.cc:91-105), then Frame::createStereo calls searchByStereo (include/ORB_SLAM2/Frame.h:316-319) -- through the C++
    drop-in classes (tests/cpp/test_dropin.cpp, mode `latency`; no Python in that number).

    r6: (b) runs FIRST, n_cpp = 2000 frames per call shape, while NO other process holds a context on the GPU -- bench.py
    starts this leg's process before it touches the GPU itself, and this function starts the C++ harness before it creates its own context.
    Until r6 the harness ran beside two idle contexts (bench.py's and this process's): that, not the library, was the p99 of 0.8 - 1.0 ms of
    the r5 record (tools/exp/latency_contexts.sh measures the same harness alone / beside one / beside two idle HIP processes)."""
    import subprocess  # noqa: F401
    import tempfile  # noqa: F401

    from orb_slam2_ros2_amd import synth
    from orb_slam2_ros2_amd._lib import Context
    from orb_slam2_ros2_amd.digest import pair_digest
    L, R = synth.stereo_pair(0, W, H)
    cpp = None
    try:
        cpp = cpp_latency_harness(n_cpp, L, R)   # before this process has a HIP context
    except (subprocess.CalledProcessError, RuntimeError, OSError) as ex:
        cpp = ex
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_v1.json")))["bench_pairs"]
    ctx = Context(W, H, NFEAT, NLEVELS, SCALE, TH_HI, TH_LO, device_id=device_id, max_images=2)
    (lk, ld), (rk, rd) = ctx.extract_batch([L, R])
    nm, ru, dp, _, _ = ctx.stereo_match(0, 1, FX, BF)
    if pair_digest(lk, ld, rk, rd, ru, dp, nm) != gold["0"]:
        raise SystemExit("bench.py: latency leg: the single-pair path differs from the golden digest")

    def one():
        ctx.extract_batch([L, R])
        ctx.stereo_match(0, 1, FX, BF)
    out = {"pair": "synthetic KITTI-shaped frame 0, 1241x376, 2000 features per image", "verified": True,
           "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "unset (runtime default: what a drop-in user has)"),
           "extract_batch_plus_match": dict(_stats_ms(one, n, warm=30), what="orbfe_extract_batch([L, R]) + orbfe_stereo_match, host to host, from Python")}
    # Frame::createStereo's device work (Frame.h:313-323: two extractions, then searchByStereo) as ONE call: the same kernels in one
    # launch sequence, one synchronisation
    (flk, fld), (frk, frd), fnm, fru, fdp = ctx.frame_stereo(L, R, FX, BF)
    if pair_digest(flk, fld, frk, frd, fru, fdp, fnm) != gold["0"]:
        raise SystemExit("bench.py: latency leg: orbfe_frame_stereo differs from the golden digest")
    out["frame_stereo_one_call"] = dict(_stats_ms(lambda: ctx.frame_stereo(L, R, FX, BF), n, warm=30),
                                        what="orbfe_frame_stereo(L, R): both extractions + the stereo match as one launch sequence, host to host, from Python")
    # BASELINE config 5's front half: one TUM-shaped RGB-D frame (640 x 480, 1000 features), host to host -- Tracking::grabFrame's cvtColor +
    # the RGB-D Frame constructor (src/Tracking.cc:55-68, src/Frame.cc:125-159): colour image -> gray -> extraction, then undistortion + the
    # depth / rightU lookup (results of both calls checked against the oracle in tests/test_frame_glue.py; here: every repetition equal)
    try:
        tum = dict(fx=520.908620, fy=521.007327, cx=325.141442, cy=249.701764, k1=0.231222, k2=-0.784899, p1=-0.003257, p2=-0.000105,
                   k3=0.917205, bf=40.0)
        rg = np.random.default_rng(5)
        g = synth.mono_image(4, 640, 480)
        bgr = np.stack([g, np.roll(g, 1, 1), np.roll(g, 2, 0)], 2).copy()
        dep = rg.integers(0, 30000, (480, 640)).astype(np.uint16)
        cr = Context(640, 480, n_features=1000, device_id=device_id, max_images=1)
        k0_, d0_ = cr.extract_color(bgr, 2)
        ku0, dd0, ru0 = cr.frame_rgbd(0, tum, dep, 5000.0)

        def rgbd_frame():
            k_, d_ = cr.extract_color(bgr, 2)
            ku, dd, ru_ = cr.frame_rgbd(0, tum, dep, 5000.0)
            return k_, d_, ku, dd, ru_
        k1_, d1_, ku1, dd1, ru1 = rgbd_frame()
        if not (np.array_equal(d0_, d1_) and ku0.tobytes() == ku1.tobytes() and np.array_equal(dd0, dd1) and np.array_equal(ru0, ru1)):
            raise SystemExit("bench.py: latency leg: the RGB-D frame is not repeatable")
        out["rgbd_frame_tum"] = dict(_stats_ms(rgbd_frame, 200, warm=20), keypoints=int(len(k0_)),
                                     what="640x480 BGR image + 16-bit depth image in, undistorted keypoints / descriptors / depth / rightU out: "
                                          "orbfe_extract_color + orbfe_frame_rgbd (BASELINE config 5's frame, 1000 features), from Python")
        # ... and as ONE call (orbfe_frame_rgbd_image: Frame::createRGBD's device work as one launch sequence; the depth image is not uploaded)
        k2_, d2_, dd2, ru2 = cr.frame_rgbd_image(bgr, tum, dep, 5000.0, 2)
        if not (np.array_equal(d2_, d0_) and k2_.tobytes() == ku0[:len(k2_)].tobytes() and np.array_equal(dd2, dd0) and np.array_equal(ru2, ru0)):
            raise SystemExit("bench.py: latency leg: orbfe_frame_rgbd_image differs from orbfe_extract_color + orbfe_frame_rgbd")
        out["rgbd_frame_tum_one_call"] = dict(_stats_ms(lambda: cr.frame_rgbd_image(bgr, tum, dep, 5000.0, 2), 200, warm=20),
                                              what="the same frame through orbfe_frame_rgbd_image, from Python")
        cr.close()
    except (RuntimeError, OSError) as ex:
        out["rgbd_frame_tum"] = {"error": f"{type(ex).__name__}: {ex}"}
    # the per-frame guided matchers of Tracking (searchByProjection x 2-4 per frame over findFeaturesInArea + getBestMatch, src/ORBMatcher.cc:
    # 265-347, 561-612; MapPoint::isInVision, src/MapPoint.cc:141-201): 1000 queries against the 2000 features of the frame just built
    r = np.random.default_rng(0)
    nq = 1000
    q = r.integers(0, len(rk), nq)
    qxy = np.stack([rk["x"][q], rk["y"][q]], 1).astype(np.float32) + r.normal(0, 3, (nq, 2)).astype(np.float32)
    rad = r.uniform(5, 40, nq).astype(np.float32)
    lo, hi = np.zeros(nq, np.int8), np.full(nq, 7, np.int8)
    a = ctx.search_in_area(0, qxy, rad, lo, hi, rd[q])
    b_ = ctx.search_in_area_features(lk, ld, qxy, rad, lo, hi, rd[q])
    if not all(np.array_equal(x, y) for x, y in zip(a, b_)):
        raise SystemExit("bench.py: latency leg: the guided search against the slot differs from the one against the uploaded features")
    pos = r.uniform(-5, 5, (2000, 3)).astype(np.float32)
    pos[:, 2] = r.uniform(3, 30, 2000)
    vd = np.tile(np.array([0, 0, 1], np.float32), (2000, 1))
    mx, mn = np.full(2000, 100, np.float32), np.full(2000, 0.1, np.float32)
    cam, bnd = (FX, FX, 607.19, 185.2), (0, W, 0, H)
    out["guided_matchers"] = {
        "what": "host arrays in, host results out, 1000 queries / 2000 map points against a 2000-feature frame (results checked against the "
                "oracle in tests/test_guided_search.py; here: slot-resident and uploaded targets agree)",
        "search_in_area_ms": _stats_ms(lambda: ctx.search_in_area(0, qxy, rad, lo, hi, rd[q]), 200, warm=10)["median_ms"],
        "search_in_area_features_ms": _stats_ms(lambda: ctx.search_in_area_features(lk, ld, qxy, rad, lo, hi, rd[q]), 200, warm=10)["median_ms"],
        "project_map_points_ms": _stats_ms(lambda: ctx.project_map_points(pos, vd, mx, mn, np.eye(3), np.zeros(3), cam, bnd), 200, warm=10)["median_ms"]}
    # Tracking::trackLocalMap's chain for 2000 local map points (Tracking.cc:641-675): searchByProjection(frame, map points, th) + OptimizePoseOnly,
    # as ONE call (orbfe_track_local_map: the frame's features are the slot's, one upload, one download) against the same three steps through
    # the separate entry points (three round trips, the queries / edges marshalled on the host in between).  The map: the left image's
    # keypoints back-projected at their stereo depth (tests/test_track_chain.py holds the fused call to the oracle's chain).
    n_l = len(lk)
    ru_full = np.full(NFEAT, -1.0)
    ru_full[:n_l] = ru[:n_l]
    depth = np.where(dp[:n_l] > 0, dp[:n_l], r.uniform(4, 30, n_l))
    CXk, CYk = 607.1928, 185.2157
    Xmp = np.stack([(lk["x"] - CXk) / FX * depth, (lk["y"] - CYk) / FX * depth, depth], 1).astype(np.float32)
    take = np.concatenate([r.permutation(n_l)[: min(n_l, 1800)], r.integers(0, n_l, 2000 - min(n_l, 1800))])
    mp_pos = Xmp[take] + r.normal(0, 0.01, (2000, 3)).astype(np.float32)
    mp_desc = ld[take].copy()
    mp_vd = (mp_pos / np.linalg.norm(mp_pos, axis=1, keepdims=True)).astype(np.float32)
    dist = np.linalg.norm(mp_pos, axis=1)
    mp_max, mp_min = (dist * 1.8).astype(np.float32), (dist * 0.6).astype(np.float32)
    mp_flags = np.full(2000, 7, np.uint8)
    sf = np.array([np.float32(SCALE) ** l for l in range(NLEVELS)], np.float32)
    sig2 = (sf * sf).astype(np.float32)
    isig2 = (np.float32(1.0) / sig2).astype(np.float32)
    Rc, tc = np.eye(3, dtype=np.float32), np.array([0.03, -0.02, 0.04], np.float32)
    p0 = np.array([0, 0, 0, 1, 0.03, -0.02, 0.04], np.float64)
    camk, bndk = (FX, FX, CXk, CYk, BF), (0.0, float(W), 0.0, float(H))

    def chain_fused():
        return ctx.track_local_map(0, mp_pos, mp_vd, mp_max, mp_min, mp_desc, mp_flags, Rc, tc, camk, bndk, p0, sig2, isig2, right_u=ru_full)

    def chain_three_calls():
        pr = ctx.project_map_points(mp_pos, mp_vd, mp_max, mp_min, Rc, tc, camk[:4], bndk)
        idx = np.flatnonzero(pr["visible"])
        lvl = pr["level"][idx].astype(np.int64)
        radius = ((np.where(pr["cos_theta"][idx] > np.float32(0.998), np.float32(2.5), np.float32(4.0)) * np.float32(3.0)) * sig2[lvl]).astype(np.float32)
        bi, bd, sd, nc = ctx.search_in_area(0, pr["uv"][idx], radius, np.maximum(0, lvl - 1).astype(np.int8), np.minimum(NLEVELS - 1, lvl + 1).astype(np.int8),
                                            mp_desc[idx])
        ok = (nc > 0) & (bd < 50) & (bd.astype(np.float32) / sd.astype(np.float32) < np.float32(0.8))
        held = np.full(NFEAT, -1, np.int64)
        for k in np.flatnonzero(ok):          # the reference's loop, map-point order (first claim wins)
            if held[bi[k]] < 0:
                held[bi[k]] = idx[k]
        ef = np.flatnonzero(held >= 0)
        meas = np.stack([lk["x"][ef].astype(np.float64), lk["y"][ef].astype(np.float64), ru_full[ef]], 1)
        oc = lk["octave"][ef]
        return ctx.pose_only_optimize(mp_pos[held[ef]].astype(np.float64), meas, isig2[oc].astype(np.float64), sig2[oc], p0, *cam32), held

    # (the camera constants as the reference holds them -- Camera::mfFx ... are floats -- so that both paths optimise the same problem to the bit:
    #  with 718.856 as a double on one side the two trajectories part after a few iterations and take different numbers of passes)
    cam32 = tuple(float(np.float32(v)) for v in (FX, FX, CXk, CYk, BF))
    gf = chain_fused()
    (ng3, pose3, _), held3 = chain_three_calls()
    if not np.array_equal(gf["assigned"], held3) or abs(gf["n_good"] - ng3) > 1 or np.abs(gf["pose"] - pose3).max() > 1e-6:
        raise SystemExit("bench.py: latency leg: the fused tracking chain differs from the three separate calls")
    out["track_local_map"] = {
        "what": "Tracking::trackLocalMap's device work for 2000 local map points against a 2000-feature frame, host arrays in, host results "
                "out: orbfe_track_local_map (one call) vs orbfe_project_map_points + orbfe_search_in_area + orbfe_pose_only_optimize with "
                "the reference's policy in numpy between them",
        "n_matches": int(gf["n_matches"]), "n_edges": int(gf["n_edges"]), "n_good": int(gf["n_good"]),
        "fused": _stats_ms(chain_fused, 200, warm=10), "three_calls": _stats_ms(chain_three_calls, 100, warm=5), "verified": True}
    # Tracking::trackMotionModel's chain (Tracking.cc:385-396): searchByProjection(frame, lastFrame, 15 [, 30]) -- a search around the last frame's
    # feature positions, last match wins -- + OptimizePoseOnly, as ONE call (orbfe_track_motion_model) against orbfe_search_in_area +
    # orbfe_pose_only_optimize with the policy in numpy.  The last frame: 1600 of the frame's keypoints a few pixels off, descriptors 6 bits off.
    qi = np.sort(r.permutation(n_l)[: min(n_l, 1600)])
    m_qxy = np.stack([lk["x"][qi], lk["y"][qi]], 1).astype(np.float32) + r.normal(0, 3, (len(qi), 2)).astype(np.float32)
    m_oct = lk["octave"][qi].astype(np.int8)
    m_lo, m_hi = np.maximum(0, m_oct - 1).astype(np.int8), np.minimum(NLEVELS - 1, m_oct + 1).astype(np.int8)
    m_desc = ld[qi].copy()
    fb = r.integers(0, 256, (len(qi), 6))
    for k in range(6):
        m_desc[np.arange(len(qi)), fb[:, k] // 8] ^= (1 << (fb[:, k] % 8)).astype(np.uint8)
    m_pos = Xmp[qi]

    def motion_fused():
        return ctx.track_motion_model(0, m_qxy, m_oct, m_lo, m_hi, m_desc, m_pos, camk, bndk, p0, sig2, isig2, right_u=ru_full)

    def motion_two_calls():
        rad15 = (np.float32(15.0) * sig2[m_oct.astype(np.int64)]).astype(np.float32)
        bi, bd, sd, nc = ctx.search_in_area(0, m_qxy, rad15, m_lo, m_hi, m_desc)
        ok = np.flatnonzero((nc > 0) & (bd < 50) & (bd.astype(np.float32) / sd.astype(np.float32) < np.float32(0.9)))
        held = np.full(NFEAT, -1, np.int64)
        held[bi[ok]] = ok                      # setMapPoints in query order: the last one stays (ascending assignment, duplicates overwritten)
        ef = np.flatnonzero(held >= 0)
        meas = np.stack([lk["x"][ef].astype(np.float64), lk["y"][ef].astype(np.float64), ru_full[ef]], 1)
        oc = lk["octave"][ef]
        return ctx.pose_only_optimize(m_pos[held[ef]].astype(np.float64), meas, isig2[oc].astype(np.float64), sig2[oc], p0, *cam32), held, len(ok)

    gm = motion_fused()
    (ngm, posem, _), heldm, nmm = motion_two_calls()
    if gm["passes"] != 1 or gm["n_matches"] != nmm or not np.array_equal(gm["assigned"], heldm) or abs(gm["n_good"] - ngm) > 1 or np.abs(gm["pose"] - posem).max() > 1e-6:
        raise SystemExit("bench.py: latency leg: the fused motion-model chain differs from the separate calls")
    out["track_motion_model"] = {
        "what": "Tracking::trackMotionModel's device work for 1600 last-frame features with map points against a 2000-feature frame, host arrays "
                "in, host results out: orbfe_track_motion_model (one call) vs orbfe_search_in_area + orbfe_pose_only_optimize with the "
                "reference's policy in numpy between them",
        "n_matches": int(gm["n_matches"]), "n_edges": int(gm["n_edges"]), "n_good": int(gm["n_good"]),
        "fused": _stats_ms(motion_fused, 200, warm=10), "two_calls": _stats_ms(motion_two_calls, 100, warm=5), "verified": True}
    ctx.close()
    # (b) the C++ drop-in: measured at the top of this function (alone on the GPU), reported here
    if isinstance(cpp, Exception):
        out["two_threads_extract_slot_plus_match"] = {"error": f"{type(cpp).__name__}: {cpp}"}
        return out
    f, q, hwq = cpp
    if int(f[8]) != len(lk) or int(f[9]) != nm:
        raise SystemExit("bench.py: latency leg: the drop-in frame differs from the verified single-pair result")

    def dist(way):
        n_, p50, p90, p99, p999, mx, e50, e99 = q[way]
        return {"median_ms": p50 / 1e3, "p90_ms": p90 / 1e3, "p99_ms": p99 / 1e3, "p999_ms": p999 / 1e3, "max_ms": mx / 1e3, "n": int(n_),
                "extract_median_ms": e50 / 1e3, "extract_p99_ms": e99 / 1e3}
    out["cpp_harness"] = ("tests/cpp/test_dropin latency, %d frames per call shape, run as the ONLY process with a context on the GPU (before this "
                          "leg's own Python measurements); every frame hashed equal to the first" % int(f[1]))
    out["two_threads_extract_slot_plus_match"] = dict(dist("two_threads"),
        what="ORB_SLAM2_ROS2::ORBExtractor x 2 on two std::threads (orbfe_extract_slot each, thread start / join included as in "
             "Frame::Frame) + searchByStereo, C++ drop-in, host cv::Mat in, std::vector<cv::KeyPoint> / cv::Mat descriptors out")
    out["same_objects_one_thread"] = dist("one_thread")
    out["createStereo_one_call_cpp"] = dict(dist("create_stereo"),
        what="the same Frame built by orbfe::dropin::createStereo (ORBExtractor::extractStereo -> orbfe_frame_stereo_slots): both "
             "extractions and the stereo match as one device call in place of the two threads and searchByStereo; every frame "
             "hashed equal to the two-thread one")
    out["two_threads_eager_start"] = dict(dist("two_threads_eager"),
        what="the reference's own shape again -- two extractor objects, two std::threads, searchByStereo -- with "
             "orbfe::ORBExtractor::eagerStart(): the constructors (which run before the threads exist, Frame.cc:91-92, and build the "
             "pyramid in the reference) enqueue the extraction (orbfe_extract_slot_begin), extract() collects it: the device works while "
             "the threads are created")
    out["cpp_hw_queues"] = hwq
    return out
