"""The reference's OWN translation units and class declarations through the compiler against the drop-in (CPU-only, in
the build container only -- /root/reference does not exist on the GPU box, where this module skips).

  1. src/ORB_SLAM2/src/Frame.cc of the reference, UNCHANGED, with include/ORB_SLAM2/ORBExtractor.h replaced by the one-line include of
     INTEGRATION.md section 2 -- `Frame::Frame` with its two extractor objects and two std::threads (Frame.cc:85-111), the RGB-D constructor,
     `Frame::createStereo` / `createRGBD` (Frame.h:313-331) and everything else of the file;
  2. tests/cpp/ref_bodies.cpp: the one-line ORBMatcher / Optimizer member bodies of INTEGRATION sections 3 and 4, the reference's call sites
     (Tracking.cc:361-396, :650-658, LocalMapping.cc:95-97) and the fused call shapes, against the reference's real Frame / KeyFrame / MapPoint /
     Map / Camera / Sim3Ret declarations with the friend line of INTEGRATION section 3 added -- and, as a control, WITHOUT that line (must fail
     on protected members: the check is live).

`g++ -std=c++17 -fsyntax-only`: nothing is linked or run.  OpenCV, DBoW3, rclcpp, g2o, Eigen and the protoc output are stand-in headers
under tests/cpp/stubs (declarations only -- this image has none of them); the reference's headers are reached through a temporary include
directory of symlinks (plus patched temporary copies for the friend line), never copied into the repo."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/ORB_SLAM2"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "include", "ORB_SLAM2")) or shutil.which("g++") is None,
                                reason="needs /root/reference (build container only) and g++")


def _include_dir(tmp, friend_line):
    """<tmp>/ORB_SLAM2/*.h: symlinks to the reference's headers, ORBExtractor.h = the one-line include, and --
    friend_line -- Frame.h / KeyFrame.h as temporary copies with `friend struct orbfe::dropin::Bodies;` beside their friend declarations"""
    d = os.path.join(tmp, "ORB_SLAM2")
    os.makedirs(d)
    src = os.path.join(REF, "include", "ORB_SLAM2")
    for f in os.listdir(src):
        if f != "ORBExtractor.h":
            os.symlink(os.path.join(src, f), os.path.join(d, f))
    with open(os.path.join(d, "ORBExtractor.h"), "w") as fh:
        fh.write("#pragma once\n#include <orbfe_dropin.hpp>      // ORB_SLAM2_ROS2::ORBExtractor on liborbfe_hip.so\n")
    if friend_line:
        for name, anchor, want in (("Frame.h", "  friend class ORBMatcher;", 2), ("KeyFrame.h", "  friend class Map;", 1)):
            txt = open(os.path.join(src, name)).read()
            assert len(re.findall("^" + re.escape(anchor) + "$", txt, re.M)) == want, f"{name}: the reference's friend declarations moved"
            os.unlink(os.path.join(d, name))
            with open(os.path.join(d, name), "w") as fh:
                fh.write(re.sub("^" + re.escape(anchor) + "$", anchor + "\n  friend struct orbfe::dropin::Bodies;", txt, flags=re.M))
    return tmp


def _syntax_only(inc, tu):
    stubs = os.path.join(ROOT, "tests", "cpp", "stubs")
    return subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I" + inc, "-I" + stubs, "-I" + os.path.join(stubs, "refgen"),
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "orb_slam2_ros2_amd", "host"), tu],
                          capture_output=True, text=True, timeout=600)


def test_reference_frame_cc_compiles_unchanged_against_the_drop_in_extractor(tmp_path):
    inc = _include_dir(str(tmp_path / "inc"), friend_line=False)
    r = _syntax_only(inc, os.path.join(REF, "src", "Frame.cc"))
    assert r.returncode == 0, r.stderr[-4000:]


def test_integration_bodies_compile_against_the_reference_classes(tmp_path):
    inc = _include_dir(str(tmp_path / "inc"), friend_line=True)
    r = _syntax_only(inc, os.path.join(ROOT, "tests", "cpp", "ref_bodies.cpp"))
    assert r.returncode == 0, r.stderr[-6000:]
    # ... and Frame.cc still compiles with the patched headers
    r = _syntax_only(inc, os.path.join(REF, "src", "Frame.cc"))
    assert r.returncode == 0, r.stderr[-4000:]


def test_without_the_friend_line_the_bodies_are_refused(tmp_path):
    """control: the bodies read protected members (mvFeatsLeft, mpExtractorLeft ...); friendship of ORBMatcher / Optimizer does not reach the
    functions they call, so without INTEGRATION's friend line the compiler must object -- which also shows that the templates ARE instantiated"""
    inc = _include_dir(str(tmp_path / "inc"), friend_line=False)
    r = _syntax_only(inc, os.path.join(ROOT, "tests", "cpp", "ref_bodies.cpp"))
    assert r.returncode != 0 and "protected within this context" in r.stderr and "mvFeatsLeft" in r.stderr
