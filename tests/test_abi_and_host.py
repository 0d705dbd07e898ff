"""Host logic that needs no GPU: the C-ABI library loads and exports every symbol include/orbfe.h declares,
fails loudly without a device, the template parser, frame sharding."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "orbfe.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(orbfe_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from orb_slam2_ros2_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 20 and set(syms) == set(_lib.EXPORTS)
    lib = ctypes.CDLL(_lib.LIB_PATH)  # loading must not need a GPU
    for s in syms:
        assert hasattr(lib, s), s
    assert lib.orbfe_abi_version() == 4


def test_no_cpu_fallback_create_fails_loudly_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    from orb_slam2_ros2_amd._lib import Context, OrbfeError
    with pytest.raises(OrbfeError) as ei:
        Context(1241, 376)
    assert ei.value.status == 3 and "no CPU fallback" in str(ei.value)
    from orb_slam2_ros2_amd import ORBExtractor
    with pytest.raises(OrbfeError):
        ORBExtractor(np.zeros((376, 1241), np.uint8), 2000, 8, 1.2)


def test_create_refuses_what_its_records_cannot_hold():
    """The argument check of orbfe_create runs before the device is looked for, so it is testable here: candidate records carry x and y in
    12 bits and the matcher's KpX record the patch centre in 14 (r6: the bound is explicit), the level table holds ORBFE_MAX_LEVELS = 16."""
    from orb_slam2_ros2_amd._lib import Context, OrbfeError
    for kw in (dict(width=4097, height=376), dict(width=1241, height=4097), dict(width=0, height=376), dict(width=1241, height=376, n_levels=17),
               dict(width=1241, height=376, n_levels=0), dict(width=1241, height=376, scale_factor=1.0), dict(width=1241, height=376, n_features=70000)):
        w, h = kw.pop("width"), kw.pop("height")
        with pytest.raises(OrbfeError) as ei:
            Context(w, h, **kw)
        assert ei.value.status == 1 and "bad config" in str(ei.value), kw     # ORBFE_EBADARG, not ORBFE_EDEVICE


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "orb_slam2_ros2_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".inc")) or fn == "Makefile":
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert "oracle" not in txt.lower(), f"{fn} mentions the oracle"


def test_brief_template_parser(tmp_path):
    from orb_slam2_ros2_amd.frontend import load_brief_template
    inc = open(os.path.join(ROOT, "orb_slam2_ros2_amd", "csrc", "brief_pattern.inc")).read()
    nums = [int(v) for v in re.findall(r"-?\d+", "\n".join(l for l in inc.split("\n") if l.lstrip().startswith("{")))]
    pat = np.asarray(nums, np.int8).reshape(256, 4)
    p = tmp_path / "brief_template.txt"
    # same shape as config/brief_template.txt: header line, tab-separated, trailing tab on some lines, no final newline
    p.write_text("x1  y1  x2  y2\n" + "\n".join("\t".join(str(v) for v in r) + ("\t" if i % 7 == 0 else "") for i, r in enumerate(pat)))
    assert np.array_equal(load_brief_template(str(p)), pat)
    assert pat[0].tolist() == [8, -3, 9, 5] and pat.min() == -13 and pat.max() == 12
    with pytest.raises(FileNotFoundError):
        load_brief_template(str(tmp_path / "missing.txt"))
    # what initBriefTemplate does with other files (ORBExtractor.cc:251-265): no count check -- a longer file acts through its first 256
    # pairs (computeBRIEF copies 32 bytes, :405-406); a line that does not parse is a pair of zeros from the failed value on; a final
    # newline adds no pair.  Fewer than 256 pairs would index past the template in the reference: refused.
    rows = ["\t".join(str(v) for v in r) for r in pat]
    p.write_text("hdr\n" + "\n".join(rows + ["1 2 3 4", "5 6 7 8"]) + "\n")
    assert np.array_equal(load_brief_template(str(p)), pat)
    broken = list(rows)
    broken[3], broken[4] = "", "7 x 9 9"
    p.write_text("hdr\n" + "\n".join(broken))
    got = load_brief_template(str(p))
    assert got[3].tolist() == [0, 0, 0, 0] and got[4].tolist() == [7, 0, 0, 0] and np.array_equal(got[5:], pat[5:])
    # operator>> takes the longest numeric PREFIX of a token: "7x" reads 7 and the next value fails on the 'x' (ADVICE r5); "2.9" truncates
    broken[4], broken[5] = "7x 9 9 9", "2.9 -2.9 3 1e1"
    p.write_text("hdr\n" + "\n".join(broken))
    got = load_brief_template(str(p))
    assert got[4].tolist() == [7, 0, 0, 0] and got[5].tolist() == [2, -2, 3, 10]
    broken[6] = "1 2 300 4"          # no int8: refused by both loaders (the C++ cast would be undefined)
    p.write_text("hdr\n" + "\n".join(broken))
    with pytest.raises(ValueError):
        load_brief_template(str(p))
    p.write_text("hdr\n" + "\n".join(rows[:255]))
    with pytest.raises(ValueError):
        load_brief_template(str(p))


def test_frame_range_partition():
    from orb_slam2_ros2_amd.sharding import frame_range
    assert [frame_range(4541, r, 8) for r in range(8)][0] == (0, 568)
    assert frame_range(4541, 7, 8) == (3976, 4541)
    for F, Wd in ((4541, 8), (100, 3), (5, 8), (0, 2)):
        cover = []
        for r in range(Wd):
            b, e = frame_range(F, r, Wd)
            cover += list(range(b, e))
        assert cover == list(range(F))


def _build_shim(tmp_path):
    import subprocess
    exe = str(tmp_path / "test_shim")
    pkg = os.path.join(ROOT, "orb_slam2_ros2_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_shim.cpp"), "-L" + pkg,
                           "-lorbfe_hip", "-pthread", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def _build_dropin(tmp_path):
    """tests/cpp/test_dropin.cpp: the cv::Mat drop-in classes and the templated Frame / KeyFrame adapters (host/orbfe_dropin.hpp) against the
    stand-in opencv2/core.hpp of tests/cpp/stubs -- a compile-and-logic check; it pins nothing about OpenCV."""
    import subprocess
    exe = str(tmp_path / "test_dropin")
    pkg = os.path.join(ROOT, "orb_slam2_ros2_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "tests", "cpp", "stubs"), "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "test_dropin.cpp"), "-L" + pkg, "-lorbfe_hip", "-pthread",
                           "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_host_mirror_compiles_and_fails_loudly_without_device(tmp_path):
    import subprocess

    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    from orb_slam2_ros2_amd import synth
    exe = _build_shim(tmp_path)
    L, R = synth.stereo_pair(0, 640, 240, n_rect=100)
    L.tofile(tmp_path / "L.raw")
    R.tofile(tmp_path / "R.raw")
    r = subprocess.run([exe, str(tmp_path / "L.raw"), str(tmp_path / "R.raw"), "640", "240"], capture_output=True, text=True)
    assert r.returncode == 3 and r.stdout.strip() == "NO_DEVICE"


def test_dropin_header_compiles_against_the_stand_in_opencv_and_fails_loudly_without_device(tmp_path):
    """host/orbfe_dropin.hpp (cv::Mat ORBExtractor, the templated bodies of searchByStereo(FramePtr), OptimizePoseOnly(FramePtr),
    OptimizeLocalMap(KeyFramePtr, bool&)) goes through g++ with -Wall against tests/cpp/stubs/opencv2/core.hpp.  A compile check of the
    adapters' use of the reference's accessors; it pins nothing about OpenCV."""
    import subprocess

    import torch
    exe = _build_dropin(tmp_path)
    # host-only mode: the write-back policy of OptimizeLocalMap at exactly 30 % outliers in a keyframe (float quotient vs double 0.3)
    pol = subprocess.run([exe, "policy"], capture_output=True, text=True)
    assert pol.returncode == 0 and pol.stdout.split() == ["POLICY_OK", "1", "0", "0", "1"], pol.stdout + pol.stderr
    # host-only: the matcher bodies instantiated on frame classes with PROTECTED data members + `friend struct orbfe::dropin::Bodies;`
    acc = subprocess.run([exe, "access"], capture_output=True, text=True)
    assert acc.returncode == 0 and acc.stdout.split() == ["ACCESS_OK", "0"], acc.stdout + acc.stderr
    if torch.cuda.is_available():
        pytest.skip("a device is present: the run-time half is tests/test_gpu_dropin.py")
    r = subprocess.run([exe, "poseonly"], capture_output=True, text=True)
    assert r.returncode == 3 and r.stdout.strip() == "NO_DEVICE"


def test_no_kernel_reads_the_dispatch_packet():
    """A kernel with a private array the compiler cannot scalarise (or with blockDim / gridDim) gets the dispatch pointer: its waves then begin
    with a scalar load from the queue's ring buffer in HOST memory (r6: 13.6 us per wave of a pair's descriptor launch).  tools/check_dispatch_ptr.py
    compiles every kernel file for gfx950 and looks at the kernel descriptors."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dispatch_ptr.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
