"""GPU tests of the drop-in layer with the reference's own signatures (orb_slam2_ros2_amd/host/orbfe_dropin.hpp) through
tests/cpp/test_dropin.cpp: the two-thread extraction pattern of Frame::Frame (src/Frame.cc:91-105) 200 times against the single-thread
result and the oracle, and the Frame / KeyFrame adapters of searchByStereo, OptimizePoseOnly, OptimizeLocalMap against the array-level
path.  cv::Mat / cv::KeyPoint come from the stand-in header tests/cpp/stubs/opencv2/core.hpp: a compile-and-logic check of the
adapters, nothing about OpenCV itself."""
import subprocess

import numpy as np
import pytest

from orb_slam2_ros2_amd import synth

pytestmark = pytest.mark.gpu

FX, BF = 718.856, 718.856 * 0.537166


def _fnv1a(b):
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    from test_abi_and_host import _build_dropin
    return _build_dropin(tmp_path_factory.mktemp("dropin"))


def test_two_extractor_objects_on_two_threads_200_times(orc, exe, tmp_path):
    L, R = synth.stereo_pair(3)
    L.tofile(tmp_path / "L.raw")
    R.tofile(tmp_path / "R.raw")
    out = subprocess.run([exe, "threads", str(tmp_path / "L.raw"), str(tmp_path / "R.raw"), "1241", "376", "200"], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    tag, iters, nl, nr, nm, hk, pyr_ok, stale_refused, levels = out.stdout.split()
    ref = orc.stereo_frame(L, R, fx=FX, bf=BF)
    assert tag == "THREADS_OK" and int(iters) == 200
    assert (int(nl), int(nr), int(nm)) == (len(ref["lk"]), len(ref["rk"]), ref["n_matches"])
    assert int(hk, 16) == _fnv1a(ref["lk"].tobytes())
    assert (int(pyr_ok), int(stale_refused), int(levels)) == (1, 1, 8)


def test_device_error_on_an_extract_thread_is_captured_and_rethrown_by_the_next_call(exe, tmp_path):
    """Frame::Frame runs extract() on bare std::threads (src/Frame.cc:100-105): an exception leaving one is std::terminate.  A failing
    extract() on a foreign thread returns empty and the error surfaces at searchByStereo (the next statement of Frame::createStereo,
    Frame.h:319); on the constructing thread it propagates at once; a later good frame is unaffected."""
    L, R = synth.stereo_pair(3)
    L.tofile(tmp_path / "L.raw")
    R.tofile(tmp_path / "R.raw")
    out = subprocess.run([exe, "threaderr", str(tmp_path / "L.raw"), str(tmp_path / "R.raw"), "1241", "376"], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr     # (std::terminate would be SIGABRT: -6)
    f = out.stdout.split()
    assert f[0] == "THREADERR_OK" and f[1:7] == ["1"] * 6 and int(f[7]) > 0


def test_create_stereo_adapter_builds_the_same_frame_as_two_threads_and_search_by_stereo(orc, exe, tmp_path):
    """mode `latency` builds the Frame three ways -- two extract() threads + searchByStereo (the reference's shape), the same calls on one
    thread, and orbfe::dropin::createStereo (one device call) -- and fails unless every frame of every way hashes equal to the first"""
    L, R = synth.stereo_pair(3)
    L.tofile(tmp_path / "L.raw")
    R.tofile(tmp_path / "R.raw")
    out = subprocess.run([exe, "latency", str(tmp_path / "L.raw"), str(tmp_path / "R.raw"), "1241", "376", "40"], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().split("\n")
    f = lines[0].split()
    assert f[0] == "LATENCY_OK" and len(f) == 15
    q = {l.split()[1]: [float(v) for v in l.split()[2:]] for l in lines[1:] if l.startswith("LATQ")}
    assert set(q) == {"one_thread", "two_threads", "create_stereo", "two_threads_eager"}
    for w_, (n_, p50, p90, p99, p999, mx, e50, e99) in q.items():   # the distribution of every call shape (the harness bench.py reports from)
        assert n_ == 40 and 0 < p50 <= p90 <= p99 <= p999 <= mx and 0 < e50 <= e99
    ref = orc.stereo_frame(L, R, fx=FX, bf=BF)
    assert (int(f[8]), int(f[9])) == (len(ref["lk"]), ref["n_matches"])
    assert float(f[11]) > 0 and float(f[13]) > 0   # createStereo; the reference shape with the constructors starting the device (eagerStart)


def test_keyframe_adapter_of_optimize_local_map_equals_the_array_path(exe, tmp_path):
    import os
    pb = open(os.path.join(os.path.dirname(__file__), "golden", "map_small.pb"), "rb").read()
    (tmp_path / "in.pb").write_bytes(pb)
    for kf in ("3", "0"):
        out = subprocess.run([exe, "localba", str(tmp_path / "in.pb"), kf], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        f = out.stdout.split()
        assert f[0] == "LOCALBA_OK" and f[1:4] == ["0", "0", "0"] and f[4] == "written=1" and f[-1] == "stop=1", out.stdout
        assert int(f[5].split("=")[1]) > 0 and int(f[6].split("=")[1]) > 0       # observations erased, points moved


def test_frame_adapter_of_optimize_pose_only_equals_the_array_path(exe):
    out = subprocess.run([exe, "poseonly"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    tag, good, good_arrays, kept, marks, pose_diff, err = out.stdout.split()
    assert tag == "POSEONLY_OK" and int(pose_diff) == 0 and int(kept) == int(good) == int(marks) and float(err) < 0.02


def test_matcher_adapters_with_the_reference_signatures_equal_the_array_level_mirrors(exe, tmp_path):
    """searchByBow(VirtualFramePtr, VirtualFramePtr, vector<DMatch>&, bool, bool), searchByProjection(frame, frame, ...),
    searchByProjection(frame, mapPoints, ...) (include/ORB_SLAM2/ORBMatcher.h:42,49,52) over stand-in Frame / MapPoint classes: matches, setMapPoints,
    addMatchInTrack counts against the array-level mirrors (oracle-checked in test_guided_wrappers.py) on the same frames and map state."""
    L, R = synth.stereo_pair(3)
    L.tofile(tmp_path / "L.raw")
    R.tofile(tmp_path / "R.raw")
    out = subprocess.run([exe, "matchers", str(tmp_path / "L.raw"), str(tmp_path / "R.raw"), "1241", "376"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    tag, n1, n2, total = out.stdout.split()
    assert tag == "MATCHERS_OK" and int(n1) > 1500 and int(n2) > 1500 and int(total) > 100


def test_fused_tracking_chain_adapter_equals_the_two_bodies(exe, tmp_path):
    """dropin::trackLocalMap -- Tracking::trackLocalMap's searchByProjection(frame, local map points, th) + OptimizePoseOnly(frame)
    (src/Tracking.cc:650-658) as ONE device call -- against the two reference-signature bodies run one after the other on a twin frame
    with twin map points: match count, assignments, counters, the optimised pose (the array-level call is held to the oracle's three-step
    chain in tests/test_track_chain.py)."""
    L, R = synth.stereo_pair(4)
    L.tofile(tmp_path / "L.raw")
    R.tofile(tmp_path / "R.raw")
    out = subprocess.run([exe, "trackchain", str(tmp_path / "L.raw"), str(tmp_path / "R.raw"), "1241", "376"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    f = out.stdout.split()
    assert f[0] == "TRACKCHAIN_OK" and int(f[2]) > 800 and abs(int(f[3]) - int(f[4])) <= 1 and float(f[7]) < 1e-5


def test_fused_motion_model_chain_adapter_equals_the_two_bodies(exe, tmp_path):
    """orbfe::dropin::trackMotionModel (one device call) against searchByProjection(frame, lastFrame, 15 [, 30]) + OptimizePoseOnly on a twin
    frame: match counts, the map points the frame holds, the optimised pose and the map points' counters; a second scenario needs the search at 30"""
    L, R = synth.stereo_pair(3)
    L.tofile(tmp_path / "L.raw")
    R.tofile(tmp_path / "R.raw")
    out = subprocess.run([exe, "motionchain", str(tmp_path / "L.raw"), str(tmp_path / "R.raw"), "1241", "376"], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    tag, n_first, passes = out.stdout.split()
    assert tag == "MOTIONCHAIN_OK" and int(n_first) > 600 and int(passes) == 2


def test_back_end_matcher_adapters_with_the_reference_signatures(exe, tmp_path):
    """searchBySim3 x2, searchForTriangulation, fuse x2 (include/ORB_SLAM2/ORBMatcher.h:55-67) over stand-in KeyFrame / MapPoint / Sim3Ret /
    Map classes on a geometrically consistent pair of keyframes (a real stereo pair: the left image at the identity, the right one a
    baseline away, map points back-projected at their stereo depth): true correspondences found inside their windows and under the
    descriptor threshold, counts = adds + replacements, addObservation / MapPoint::replace called as processFuseMps prescribes, the
    epipolar bound of a rectified pair.  The device core underneath is held to the oracle in test_guided_search.py / test_matcher_ext.py."""
    L, R = synth.stereo_pair(5)
    L.tofile(tmp_path / "L.raw")
    R.tofile(tmp_path / "R.raw")
    out = subprocess.run([exe, "backend", str(tmp_path / "L.raw"), str(tmp_path / "R.raw"), "1241", "376"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    f = out.stdout.split()
    assert f[0] == "BACKEND_OK" and int(f[3]) > 200 and int(f[4]) > 200 and int(f[5]) > 50 and int(f[7]) > 100


def test_rgbd_frame_tail_adapter_equals_the_array_level_call(exe, tmp_path):
    """Frame::Frame for RGB-D input (src/Frame.cc:125-159) after extract(): undistortion + depth / rightU lookup through the adapter against
    orbfe_frame_rgbd on the same slot (which tests/test_frame_glue.py holds to the oracle)."""
    img = synth.mono_image(0)
    img.tofile(tmp_path / "G.raw")
    out = subprocess.run([exe, "rgbd", str(tmp_path / "G.raw"), "640", "480"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.split()[0] == "RGBD_OK", out.stdout + out.stderr


def test_matcher_calls_from_a_third_thread_beside_two_extracting_threads(tmp_path):
    """The reference runs fuse / searchBySim3 / searchForTriangulation on the LocalMapping and LoopClosing threads while Tracking builds
    Frames on two extractor threads (System.cc:119-129).  One context, three threads: two orbfe_extract_slot lanes and a stream of
    orbfe_search_in_area_features calls; every result must equal the single-threaded one (the per-context API lock)."""
    import threading

    import numpy as np

    from orb_slam2_ros2_amd._lib import Context
    L, R = synth.stereo_pair(5)
    ctx = Context(1241, 376, max_images=4)
    (lk, ld), (rk, rd) = ctx.extract_batch([L, R])
    r = np.random.default_rng(0)
    nq = 200
    q = r.integers(0, len(rk), nq)
    qxy = np.stack([rk["x"][q], rk["y"][q]], 1).astype(np.float32)
    rad = r.uniform(5, 60, nq).astype(np.float32)
    lo, hi = np.zeros(nq, np.int8), np.full(nq, 7, np.int8)
    want = ctx.search_in_area_features(lk, ld, qxy, rad, lo, hi, rd[q])
    errs, stop = [], threading.Event()

    def extractor(slot, img, ref_k, ref_d):
        try:
            for _ in range(150):
                k, d = ctx.extract_slot(slot, img)
                if not (np.array_equal(k, ref_k) and np.array_equal(d, ref_d)):
                    errs.append(f"slot {slot}: features differ")
                    return
        except Exception as ex:  # noqa: BLE001
            errs.append(repr(ex))

    def matcher():
        try:
            n = 0
            while not stop.is_set() or n < 20:
                got = ctx.search_in_area_features(lk, ld, qxy, rad, lo, hi, rd[q])
                if not all(np.array_equal(a, b) for a, b in zip(got, want)):
                    errs.append("search differs")
                    return
                bi, bd, sd = ctx.match_bruteforce(rd[q], ld)   # another entry point that shares the context's scratch buffer
                if bi.shape[0] != nq:
                    errs.append("bruteforce shape")
                    return
                n += 1
        except Exception as ex:  # noqa: BLE001
            errs.append(repr(ex))

    ts = [threading.Thread(target=extractor, args=(2, L, lk, ld)), threading.Thread(target=extractor, args=(3, R, rk, rd)),
          threading.Thread(target=matcher)]
    for t in ts:
        t.start()
    ts[0].join()
    ts[1].join()
    stop.set()
    ts[2].join()
    ctx.close()
    assert not errs, errs
