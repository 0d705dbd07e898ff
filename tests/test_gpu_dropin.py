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
