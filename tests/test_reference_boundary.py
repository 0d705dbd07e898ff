"""The drop-in bodies (orb_slam2_ros2_amd/host/orbfe_dropin.hpp) against the REFERENCE's own class declarations.

The drop-in is compiled here only against stand-in Frame / KeyFrame / MapPoint / Camera classes (tests/cpp/test_dropin.cpp, namespace
ref): OpenCV, g2o, DBoW3 and rclcpp are not in this image, so the reference's headers cannot go through a compiler.  This test reads
them as text instead (tools/cpp_decls.py) and checks, member by member, that

  * every accessor / field the bodies touch through a frame, key-frame, map-point or camera object is DECLARED by the reference
    (include/ORB_SLAM2/Frame.h:20-345, KeyFrame.h, MapPoint.h:141-201 + :333, Camera.h) with a parameter count the call site fits;
  * what is not public there belongs to a class that carries the friend line INTEGRATION.md prescribes (VirtualFrame / Frame / KeyFrame:
    `friend struct orbfe::dropin::Bodies;`) -- nothing private of MapPoint or Camera is touched;
  * the member functions the bodies replace exist with the parameter counts INTEGRATION.md's one-line wrappers assume
    (ORBMatcher.h:32-80, Optimizer.h:69-72, ORBExtractor.h:100-160);
  * the stand-ins do not offer the bodies anything the reference lacks: a stand-in member the reference does not declare is never named
    in orbfe_dropin.hpp, and stand-in methods that carry a reference name accept the reference's parameter counts.

CPU only, skipped where /root/reference does not exist (the GPU box)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/ORB_SLAM2/include/ORB_SLAM2"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference's headers are not on this machine")

FRAME_VARS = {"self", "pFrame", "pFrame1", "pFrame2", "pframe", "pCurr", "pLast", "pKframe", "pkframe", "pkf", "pkf1", "pkf2", "kf", "f",
              "mpCurr", "mpMatch", "item.first"}
MP_VARS = {"pMp", "pMp1", "pMp2", "pMpInF", "p", "h", "cur", "matchPMp", "era.first"}
NOT_REFERENCE_OBJECTS = {"it", "excludedHits", "exclude"}   # std iterators / vectors


@pytest.fixture(scope="module")
def ref():
    from tools.cpp_decls import parse_classes, resolve
    classes = {}
    for h in ("Frame", "KeyFrame", "MapPoint", "Camera", "ORBMatcher", "Optimizer", "ORBExtractor"):
        classes.update(parse_classes(open(os.path.join(REF, h + ".h"), encoding="utf-8", errors="ignore").read()))
    fam = {}
    for c in ("VirtualFrame", "Frame", "KeyFrame"):
        for name, ms in resolve(classes, c).items():
            fam.setdefault(name, []).extend((c, m) for m in ms)
    return {"classes": classes, "frame": fam, "mp": {n: [("MapPoint", m) for m in ms] for n, ms in classes["MapPoint"].members.items()},
            "camera": classes["Camera"].members}


def _dropin_text():
    from tools.cpp_decls import strip_comments
    return strip_comments(open(os.path.join(ROOT, "orb_slam2_ros2_amd", "host", "orbfe_dropin.hpp")).read())


def _uses(text):
    """(object expression, member, argument count or None for a field access) for every `obj->member` in the bodies"""
    from tools.cpp_decls import _match_brace, split_top
    out = []
    for m in re.finditer(r"((?:[A-Za-z_]\w*\.)?[A-Za-z_]\w*)->([A-Za-z_]\w*)\s*(\()?", text):
        obj, name, call = m.group(1), m.group(2), m.group(3)
        nargs = None
        if call:
            po = m.end() - 1
            pc = _match_brace(text, po, "(", ")")
            nargs = len([a for a in split_top(text[po + 1:pc]) if a.strip()])
        out.append((obj, name, nargs))
    return out


def test_reference_headers_are_read_as_expected(ref):
    """the reader itself, on facts of the reference's headers that are easy to see by eye"""
    c = ref["classes"]
    assert c["Frame"].bases == ["VirtualFrame"] and c["KeyFrame"].bases == ["VirtualFrame"]
    assert "ORBMatcher" in c["VirtualFrame"].friends and {"ORBMatcher", "Optimizer"} <= set(c["Frame"].friends)
    vf = c["VirtualFrame"].members
    assert vf["mvFeatsLeft"][0].access == "protected" and "std::vector<cv::KeyPoint>" in vf["mvFeatsLeft"][0].decl
    assert vf["mfMaxU"][0].access == "public" and vf["mvfScaledFactors"][0].static
    assert sorted(m.arity for m in vf["getRightU"]) == [(0, 0), (1, 1)] and vf["getScaledFactor2"][0].static
    assert c["Frame"].members["mpExtractorLeft"][0].access == "private"
    mp = c["MapPoint"].members
    assert mp["eraseObservetion"][0].arity == (1, 2) and mp["isInVision"][0].arity == (4, 4) and mp["replace"][0].static
    assert all(m.static and m.access == "public" for m in c["Camera"].members["mfFx"])


def test_every_member_the_bodies_touch_is_declared_by_the_reference(ref):
    problems, seen_nonpublic = [], set()
    for obj, name, nargs in _uses(_dropin_text()):
        base = obj.split(".")[-1] if obj not in FRAME_VARS | MP_VARS else obj
        if obj in NOT_REFERENCE_OBJECTS or base in NOT_REFERENCE_OBJECTS:
            continue
        if obj in FRAME_VARS:
            fams = ["frame"]
        elif obj in MP_VARS:
            fams = ["mp"]
        else:
            problems.append(f"{obj}->{name}: object of unknown kind (add it to FRAME_VARS / MP_VARS / NOT_REFERENCE_OBJECTS)")
            continue
        cands = [cm for f in fams for cm in ref[f].get(name, [])]
        if not cands:
            problems.append(f"{obj}->{name}: not declared by the reference's {'/'.join(fams)} classes")
            continue
        if nargs is None:
            ok = [cm for cm in cands if cm[1].kind == "field"]
            if not ok:
                problems.append(f"{obj}->{name}: used as a field, declared as {cands[0][1].kind}")
                continue
        else:
            ok = [cm for cm in cands if cm[1].kind == "method" and cm[1].arity[0] <= nargs <= cm[1].arity[1]]
            if not ok:
                problems.append(f"{obj}->{name}({nargs} arguments): the reference declares {[cm[1].decl for cm in cands]}")
                continue
        if all(cm[1].access != "public" for cm in ok):
            seen_nonpublic.add((ok[0][0], name))
            if ok[0][0] == "MapPoint":
                problems.append(f"{obj}->{name}: {ok[0][1].access} in MapPoint (no friend line is prescribed for MapPoint)")
    assert not problems, "\n".join(problems)
    # the protected / private members really are what the friend line is for (the list INTEGRATION.md names)
    names = {n for _, n in seen_nonpublic}
    assert {"mvFeatsLeft", "mvLeftDescriptor", "mvpMapPoints", "mpExtractorLeft"} <= names
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "friend struct orbfe::dropin::Bodies;" in integ


def test_camera_statics_the_bodies_read_exist(ref):
    used = set(re.findall(r"CameraT::(\w+)", _dropin_text()))
    assert used, "the bodies read the camera through CameraT::"
    for name in used:
        ms = ref["camera"].get(name)
        assert ms and ms[0].static and ms[0].access == "public" and ms[0].kind == "field", f"Camera::{name}"
    for name in ("mfFx", "mfFy", "mfCx", "mfCy", "mfBf", "mfBl"):
        assert "float" in ref["camera"][name][0].decl
    assert "cv::Mat" in ref["camera"]["mDistCoeff"][0].decl and "cv::Mat" in ref["camera"]["mKInv"][0].decl
    # the two statics called through a class template parameter
    c = ref["classes"]
    assert any(m.static and m.arity == (1, 1) for m in c["KeyFrame"].members["updateConnections"])
    assert any(m.static and m.arity == (3, 3) for m in c["MapPoint"].members["replace"])


def test_replaced_member_functions_have_the_parameter_counts_the_wrappers_assume(ref):
    c = ref["classes"]
    om, op, ex = c["ORBMatcher"].members, c["Optimizer"].members, c["ORBExtractor"].members

    def has(members, name, n):
        return any(m.kind == "method" and m.arity[0] <= n <= m.arity[1] for m in members.get(name, []))
    assert has(om, "searchByStereo", 1) and has(om, "searchByBow", 5) and has(om, "descDistance", 2)
    assert has(om, "searchByProjection", 4) and has(om, "searchBySim3", 5) and has(om, "searchForTriangulation", 3)
    assert has(om, "fuse", 3) and has(om, "getBestMatch", 5) or has(om, "getBestMatch", 4)
    assert has(op, "OptimizePoseOnly", 1) and has(op, "OptimizeLocalMap", 2)
    assert all(m.static for m in op["OptimizePoseOnly"]) and all(m.static for m in op["OptimizeLocalMap"])
    assert has(ex, "ORBExtractor", 7) and has(ex, "extract", 2) and has(ex, "getPyramid", 0) and has(ex, "getScaledFactors", 0)
    for name in ("mnLevels", "mfScaledFactor", "mvfScaledFactors", "mnBorderSize"):
        assert ex[name][0].static, name


def test_stand_ins_offer_the_bodies_nothing_the_reference_lacks(ref):
    from tools.cpp_decls import parse_classes, strip_comments
    src = open(os.path.join(ROOT, "tests", "cpp", "test_dropin.cpp")).read()
    ns = src[src.index("namespace ref {"):src.index("}  // namespace ref")]
    stand = parse_classes(ns)
    assert {"MapPoint", "VirtualFrame", "Frame", "KeyFrame", "Camera"} <= set(stand)
    body_names = set(re.findall(r"(?:->|\.|::)([A-Za-z_]\w*)", _dropin_text()))
    refsets = {"MapPoint": ref["mp"], "VirtualFrame": ref["frame"], "Frame": ref["frame"], "KeyFrame": ref["frame"],
               "Camera": {n: [("Camera", m) for m in ms] for n, ms in ref["camera"].items()}}
    problems = []
    for cname, rs in refsets.items():
        for name, ms in stand[cname].members.items():
            if name == cname or name == "~" + cname:
                continue
            if name not in rs:
                # test-only storage / counters of the stand-in: the bodies must not know about them
                if name in body_names:
                    problems.append(f"ref::{cname}::{name} is not in the reference, yet orbfe_dropin.hpp names it")
                continue
            for m in ms:
                if m.kind != "method":
                    continue
                want = [cm[1].arity for cm in rs[name] if cm[1].kind == "method"]
                if want and not any(w[1] >= m.arity[0] and w[0] <= m.arity[1] for w in want):
                    problems.append(f"ref::{cname}::{name} takes {m.arity} parameters, the reference's {want}")
    assert not problems, "\n".join(problems)
