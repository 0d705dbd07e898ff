"""map.pb boundary: host/map_pb.hpp's wire format against the real protobuf runtime, the local-map graph against a
numpy restatement of src/Optimizer.cc:232-330, and -- on the GPU -- Optimizer::OptimizeLocalMap on a map file."""
import numpy as np
import pytest

import map_pb_util as U
from orb_slam2_ros2_amd import _lib

MD = U.messages()["MapData"]


@pytest.fixture(scope="module")
def small_map():
    md, cam = U.synth_map(n_kf=15, n_pt=500, kf_id_step=2)
    return md, cam, md.SerializeToString()


def test_empty_and_malformed_inputs():
    assert _lib.map_pb_summary(b"")["n_keyframes"] == 0
    canon = _lib.map_pb_reencode(b"")  # the reference always touches mutable_keyframes() / mutable_mappoints(): both present, empty
    assert canon == b"\x0a\x00\x12\x00"
    m = MD()
    m.ParseFromString(canon)
    assert m.HasField("keyframes") and m.HasField("mappoints")
    for bad in (b"\x0a\x05\x08", b"\x0a\xff\xff\xff\xff\xff\xff\xff\xff\xff\xff\x01", b"\x0b\x00", b"\x00\x00"):
        with pytest.raises(RuntimeError):
            _lib.map_pb_summary(bad)


def test_reencode_is_what_libprotobuf_writes(small_map):
    md, _, pb = small_map
    s = _lib.map_pb_summary(pb)
    assert s["n_keyframes"] == len(md.keyframes.keyframes) and s["n_mappoints"] == len(md.mappoints.mappoints)
    assert s["n_keypoints"] == sum(len(k.keypoints) for k in md.keyframes.keyframes)
    assert s["n_observations"] == sum(sum(1 for m in k.map_points if m >= 0) for k in md.keyframes.keyframes)
    assert s["next_id"] == md.keyframes.next_id and s["n_scale_factors"] == 8
    out = _lib.map_pb_reencode(pb)
    back = MD()
    back.ParseFromString(out)
    assert back == md  # every field of every message survived
    assert len(out) == len(pb)  # same encoding up to the (unspecified) order of the BowVector map entries
    assert _lib.map_pb_reencode(out) == out  # idempotent
    # byte-exact once the map order cannot differ (at most one word per keyframe)
    md1 = MD()
    md1.CopyFrom(md)
    for k in md1.keyframes.keyframes:
        words = sorted(k.bow_vector.words.items())[:1]
        k.bow_vector.words.clear()
        for w, v in words:
            k.bow_vector.words[w] = v
    pb1 = md1.SerializeToString()
    assert _lib.map_pb_reencode(pb1) == pb1


def test_unusual_encodings_are_accepted():
    """unpacked repeated scalars, unknown fields, fields out of order, negative octave, -0.0, a short descriptor"""
    def varint(v):
        v &= (1 << 64) - 1
        out = bytearray()
        while v >= 0x80:
            out.append((v & 0x7f) | 0x80)
            v >>= 7
        out.append(v)
        return bytes(out)

    def ld(field, payload):
        return varint(field << 3 | 2) + varint(len(payload)) + payload

    f32 = lambda x: np.float32(x).tobytes()
    kp = b"\x25" + f32(-0.0) + b"\x18" + varint(-3) + b"\x0d" + f32(7.5)              # angle, octave, x (reverse order)
    kf = (varint(16 << 3 | 0) + varint(-1) + varint(16 << 3 | 0) + varint(41)          # map_points, one element per tag
          + b"\x3d" + f32(12.25) + b"\x3d" + f32(-1.0)                                 # right_u unpacked (field 7, wire type 5)
          + ld(6, kp) + ld(9, ld(1, b"\x01\x02\x03")) + varint(99 << 3 | 0) + varint(5)  # short descriptor, unknown field 99
          + ld(100, b"junk") + b"\x08" + varint(6))                                    # unknown length-delimited field, id last
    pb = ld(1, b"\x08\x07" + ld(3, kf)) + ld(7, b"zz")
    out = _lib.map_pb_reencode(pb)
    m = MD()
    m.ParseFromString(out)
    k = m.keyframes.keyframes[0]
    assert m.keyframes.next_id == 7 and k.id == 6 and list(k.map_points) == [-1, 41] and list(k.right_u) == [12.25, -1.0]
    assert k.keypoints[0].x == 7.5 and k.keypoints[0].octave == -3 and np.signbit(np.float32(k.keypoints[0].angle))
    assert k.descriptors[0].data == b"\x01\x02\x03"
    ref = MD()
    ref.ParseFromString(pb)  # libprotobuf keeps the unknown fields in its own re-serialisation; compare the known content
    assert ref.keyframes.keyframes[0].keypoints[0] == k.keypoints[0] and list(ref.keyframes.keyframes[0].map_points) == [-1, 41]
    assert k.HasField("pose") and k.HasField("bow_vector") and m.HasField("mappoints")  # mutable_*() fields always written


@pytest.mark.parametrize("kf_id", [10, 0, 28])
def test_local_graph_matches_restatement(small_map, kf_id):
    md, _, pb = small_map
    g, r = _lib.map_local_graph(pb, kf_id), U.local_graph(md, kf_id)
    assert g["n_group"] == r["n_group"] and g["n_group"] >= 10
    for key, want in r.items():
        if key == "n_group":
            continue
        got = np.asarray(g[key])
        assert got.shape == np.asarray(want).shape, key
        if got.dtype.kind == "f":
            assert np.abs(got - want).max() < 1e-15, key  # poses: sqrt / divide order of the quaternion conversion
        else:
            assert (got == want).all(), key
    # keyframe 0 is fixed inside the free group; every vertex behind the group is a fixed observer
    ids, fixed = g["pose_kf_id"], g["pose_fixed"]
    assert all(fixed[i] == (1 if (i >= g["n_group"] or ids[i] == 0) else 0) for i in range(len(ids)))
    assert np.allclose(np.linalg.norm(g["poses"][:, :4], axis=1), 1.0) and (g["poses"][:, 3] >= 0).all()


def test_local_graph_edge_cases(small_map):
    md, _, pb = small_map
    with pytest.raises(RuntimeError):
        _lib.map_local_graph(pb, 7)  # no such keyframe
    lone = MD()
    lone.CopyFrom(md)
    for k in lone.keyframes.keyframes:
        del k.connected_kfs[:]
    g = _lib.map_local_graph(lone.SerializeToString(), 10)
    assert g["n_group"] == 1 and g["pose_kf_id"][0] == 10 and g["pose_fixed"][0] == 0 and g["pose_fixed"][1:].all()
    # a map point id that the file does not hold is skipped (the reference would dereference a null pointer)
    broken = MD()
    broken.CopyFrom(md)
    broken.keyframes.keyframes[5].map_points[0] = 999999
    gb, g0 = _lib.map_local_graph(broken.SerializeToString(), 10), _lib.map_local_graph(pb, 10)
    assert len(gb["edge_pose"]) == len(g0["edge_pose"]) - 1


# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_local_ba_on_map_file(small_map):
    from oracle import pyoracle
    from orb_slam2_ros2_amd._lib import Context
    md, cam, pb = small_map
    ctx = Context(640, 480, 1000, 8, 1.2, 20, 7, max_images=2)
    kf_id = 10
    out_pb, rep = ctx.map_local_ba(pb, kf_id, **cam)
    g = _lib.map_local_graph(pb, kf_id)
    assert (rep["n_poses"], rep["n_group"], rep["n_points"], rep["n_edges"]) == (len(g["pose_kf_id"]), g["n_group"], len(g["point_id"]),
                                                                                  len(g["edge_pose"]))
    assert rep["written"] == 1 and rep["chi2_after"] < rep["chi2_before"] and rep["iterations"][0] == 5

    # the same solve by hand through the C-ABI, and by the CPU restatement of g2o's Levenberg loop
    prob = dict(g, **{k: float(np.float32(v)) for k, v in cam.items()})  # Camera::mfFx .. mfBf are floats
    dev = ctx.ba_local_optimize(prob, g["pose_fixed"])
    orc = pyoracle.Oracle(pyoracle.build()).ba_local_optimize(prob, g["pose_fixed"])
    assert np.abs(dev["poses"] - orc["poses"]).max() < 1e-7 and np.abs(dev["points"] - orc["points"]).max() < 1e-6
    assert (dev["bad"] == orc["bad"]).all()
    assert rep["n_outlier_edges"] == int(dev["bad"].sum()) and rep["n_outlier_edges"] > 0

    new = MD()
    new.ParseFromString(out_pb)
    kfs_old = {int(k.id): k for k in md.keyframes.keyframes}
    kfs_new = {int(k.id): k for k in new.keyframes.keyframes}
    # poses of the free group = Converter::ConvertSE32Tcw of the optimised estimates, as float
    for v in range(g["n_group"]):
        kid = int(g["pose_kf_id"][v])
        R = U.quat_to_R(dev["poses"][v, :4]).astype(np.float32)
        assert np.abs(np.array(kfs_new[kid].pose.rotation, np.float32).reshape(3, 3) - R).max() <= 1.2e-7
        assert (np.array(kfs_new[kid].pose.translation, np.float32) == dev["poses"][v, 4:].astype(np.float32)).all()
    assert g["n_group"] == len(g["pose_kf_id"]) or all(
        kfs_new[int(k)].pose == kfs_old[int(k)].pose for k in g["pose_kf_id"][g["n_group"]:])
    mps_new = {int(m.id): m for m in new.mappoints.mappoints}
    for p, pid in enumerate(g["point_id"]):
        m = mps_new[int(pid)]
        assert (np.array([m.position.x, m.position.y, m.position.z], np.float32) == dev["points"][p].astype(np.float32)).all()
    # exactly the outlier observations were erased; nothing else changed
    erased = {(int(g["pose_kf_id"][g["edge_pose"][e]]), int(g["edge_feat"][e])) for e in np.nonzero(dev["bad"])[0]}
    for kid, k in kfs_new.items():
        old = kfs_old[kid]
        for idx, (a, b) in enumerate(zip(old.map_points, k.map_points)):
            assert b == (-1 if (kid, idx) in erased else a)
        assert k.keypoints == old.keypoints and k.descriptors == old.descriptors and k.bow_vector == old.bow_vector
        assert list(k.right_u) == list(old.right_u) and k.connected_kfs == old.connected_kfs
    # the surviving observations sit at noise level (chi2 of a 2/3-dof residual with unit-variance noise)
    keep = dev["bad"] == 0
    assert dev["chi2"][keep].mean() < 3.0 and dev["chi2"][keep].max() <= 7.815
    ctx.close()


@pytest.mark.gpu
def test_local_ba_policy_keeps_the_map_when_too_much_would_be_erased(small_map):
    """src/Optimizer.cc:391-404: > 20 % of the affected keyframes losing > 30 % of their points => nothing is written"""
    from orb_slam2_ros2_amd._lib import Context
    md, cam, _ = small_map
    wrecked = MD()
    wrecked.CopyFrom(md)
    rng = np.random.default_rng(3)
    for k in wrecked.keyframes.keyframes:  # gross errors on most observations of every keyframe
        for i, kp in enumerate(k.keypoints):
            if k.map_points[i] >= 0 and rng.uniform() < 0.7:
                kp.x += float(np.float32(rng.uniform(60, 120)))
    pb = wrecked.SerializeToString()
    ctx = Context(640, 480, 1000, 8, 1.2, 20, 7, max_images=2)
    out_pb, rep = ctx.map_local_ba(pb, 10, **cam)
    assert rep["written"] == 0 and rep["n_bad_keyframes"] > 0.2 * rep["n_keyframes_hit"]
    same = MD()
    same.ParseFromString(out_pb)
    assert same == wrecked
    ctx.close()


@pytest.mark.gpu
def test_cpp_shim_runs_local_ba_on_a_map_file(small_map, tmp_path):
    """orbfe::Optimizer::OptimizeLocalMap(ctx, MapRec&, kfId, cam) of host/orbfe_shim.hpp writes the same file as the C-ABI call"""
    import subprocess

    from orb_slam2_ros2_amd import synth
    from orb_slam2_ros2_amd._lib import Context
    from orb_slam2_ros2_amd.ba_synth import BF, CX, CY, FX, FY
    from test_abi_and_host import _build_shim
    _, _, pb = small_map
    exe = _build_shim(tmp_path)
    L, R = synth.stereo_pair(0, 640, 240, n_rect=100)
    L.tofile(tmp_path / "L.raw")
    R.tofile(tmp_path / "R.raw")
    (tmp_path / "in.pb").write_bytes(pb)
    out = subprocess.run([exe, str(tmp_path / "L.raw"), str(tmp_path / "R.raw"), "640", "240", str(tmp_path / "in.pb"), "10",
                          str(tmp_path / "out.pb")], capture_output=True, text=True, check=True)
    n_group, n_points, n_edges, n_out, written = map(int, out.stdout.split()[-1].split("/"))
    ctx = Context(640, 480, 1000, 8, 1.2, 20, 7, max_images=2)
    want, rep = ctx.map_local_ba(pb, 10, FX, FY, CX, CY, BF)
    ctx.close()
    assert (n_group, n_points, n_edges, n_out, written) == (rep["n_group"], rep["n_points"], rep["n_edges"], rep["n_outlier_edges"], 1)
    assert (tmp_path / "out.pb").read_bytes() == want


def test_c_abi_argument_checks_of_the_host_only_entry_points(small_map):
    """NULL / size-query conventions of orbfe_map_pb_* and orbfe_map_local_graph (no device needed)."""
    import ctypes as C
    L = _lib.load()
    _, _, pb = small_map
    n = C.c_size_t(0)
    assert L.orbfe_map_pb_summary(pb, len(pb), None) == 1                      # ORBFE_EBADARG
    assert L.orbfe_map_pb_reencode(pb, len(pb), None, 0, None) == 1
    assert L.orbfe_map_pb_reencode(pb, len(pb), None, 0, C.byref(n)) == 0 and n.value == len(pb)   # size query
    small = (C.c_uint8 * 16)()
    assert L.orbfe_map_pb_reencode(pb, len(pb), small, 16, C.byref(n)) == 4 and n.value == len(pb)  # ORBFE_ECAPACITY, size reported
    sizes = (C.c_int32 * 4)()
    assert L.orbfe_map_local_graph(pb, len(pb), 10, None, None) == 1
    assert L.orbfe_map_local_graph(pb, len(pb), 10, C.byref(sizes), None) == 0 and sizes[0] >= sizes[1] > 0 and sizes[3] > 0
    assert L.orbfe_map_local_graph(pb, len(pb), 11, C.byref(sizes), None) == 1  # no such keyframe
    assert L.orbfe_map_local_graph(pb[: len(pb) // 2], len(pb) // 2, 10, C.byref(sizes), None) == 1  # truncated file
    # the device entry point refuses NULL context / camera before touching anything
    assert L.orbfe_map_local_ba(None, pb, len(pb), 10, None, None, None, 0, C.byref(n), None) == 1


# ---- the TEXT map format (Map::saveToTxtFile / loadFromTxtFile, src/Map.cc:82-165) ----------------------------------------------------
def _g(v):
    return "%g" % v            # what `std::ostream << float / double` prints by default: 6 significant digits


def _txt_of(md):
    """KeyFrames.txt / MapPoints.txt written line by line after operator<<(KeyFrame) (src/KeyFrame.cc:400-530) and operator<<(MapPoint)
    (src/MapPoint.cc:538-565) from the protobuf message"""
    k_out = [str(md.keyframes.next_id) + " " + "".join(_g(s) + " " for s in md.keyframes.scale_factors)]
    for k in md.keyframes.keyframes:
        k_out.append(f"{k.id} {_g(k.max_u)} {_g(k.max_v)} {_g(k.min_u)} {_g(k.min_v)}")
        k_out.append("".join(f"{_g(kp.x)} {_g(kp.y)} {kp.octave} {_g(kp.angle)} {_g(k.right_u[i])} {_g(k.depths[i])} " for i, kp in enumerate(k.keypoints)))
        k_out.append("".join(f"{b} " for d in k.descriptors for b in d.data))
        k_out.append("".join(f"{w} {_g(v)} " for w, v in sorted(k.bow_vector.words.items())))
        k_out.append("".join(f"{n.node_id} {len(n.feature_ids)} " + "".join(f"{i} " for i in n.feature_ids) for n in k.feature_vector.nodes))
        k_out.append("".join(_g(v) + " " for v in list(k.pose.rotation) + list(k.pose.translation)))
        k_out.append("".join(f"{c.id} {c.weight} " for c in sorted(k.connected_kfs, key=lambda c: c.id)))
        k_out.append("".join(f"{c} " for c in k.children_ids))
        k_out.append("".join(f"{c} " for c in k.loop_edges))
        k_out.append("".join(f"{m} " for m in k.map_points))
    m_out = []
    for m in md.mappoints.mappoints:
        m_out.append(f"{m.id} {_g(m.max_distance)} {_g(m.min_distance)} {m.ref_kf_id} {m.ref_feat_id} {m.matches_in_track} {m.inliers_in_track}")
        m_out.append(f"{_g(m.position.x)} {_g(m.position.y)} {_g(m.position.z)} {_g(m.view_direction.x)} {_g(m.view_direction.y)} {_g(m.view_direction.z)}")
        m_out.append("".join(f"{b} " for b in m.desc.data))
    return "\n".join(k_out) + "\n", ("\n".join(m_out) + "\n" if m_out else "")


def test_text_map_is_what_the_reference_writes_and_reads_back(small_map):
    md, _, pb = small_map
    kf_txt, mp_txt = _lib.map_pb_to_txt(pb)
    want_k, want_m = _txt_of(md)
    assert kf_txt == want_k and mp_txt == want_m
    assert kf_txt.count("\n") == 1 + 10 * len(md.keyframes.keyframes) and mp_txt.count("\n") == 3 * len(md.mappoints.mappoints)
    # reading it back (loadFromTxtFile): every number is what the 6-digit text says, as float32; integers and descriptors are exact
    back = MD()
    back.ParseFromString(_lib.map_txt_to_pb(kf_txt, mp_txt))
    assert back.keyframes.next_id == md.keyframes.next_id and len(back.keyframes.keyframes) == len(md.keyframes.keyframes)
    f32 = lambda v: float(np.float32(float(_g(v))))
    for a, b in zip(md.keyframes.keyframes, back.keyframes.keyframes):
        assert a.id == b.id and list(a.map_points) == list(b.map_points) and list(a.children_ids) == list(b.children_ids)
        assert list(a.loop_edges) == list(b.loop_edges) and [d.data for d in a.descriptors] == [d.data for d in b.descriptors]
        assert [(kp.octave, f32(kp.x), f32(kp.y), f32(kp.angle)) for kp in a.keypoints] == [(kp.octave, kp.x, kp.y, kp.angle) for kp in b.keypoints]
        assert [f32(v) for v in a.right_u] == list(b.right_u) and [f32(v) for v in a.depths] == list(b.depths)
        assert [f32(v) for v in a.pose.rotation] == list(b.pose.rotation) and [f32(v) for v in a.pose.translation] == list(b.pose.translation)
        assert sorted((c.id, c.weight) for c in a.connected_kfs) == [(c.id, c.weight) for c in b.connected_kfs]
        assert {w: float(_g(v)) for w, v in a.bow_vector.words.items()} == dict(b.bow_vector.words)
        assert [(n.node_id, list(n.feature_ids)) for n in a.feature_vector.nodes] == [(n.node_id, list(n.feature_ids)) for n in b.feature_vector.nodes]
    for a, b in zip(md.mappoints.mappoints, back.mappoints.mappoints):
        assert (a.id, a.ref_kf_id, a.ref_feat_id, a.matches_in_track, a.inliers_in_track, a.desc.data) == \
               (b.id, b.ref_kf_id, b.ref_feat_id, b.matches_in_track, b.inliers_in_track, b.desc.data)
        assert (f32(a.position.x), f32(a.position.y), f32(a.position.z), f32(a.max_distance)) == (b.position.x, b.position.y, b.position.z, b.max_distance)
    # text -> pb -> text is a fixed point (the second text form loses nothing more)
    k2, m2 = _lib.map_pb_to_txt(_lib.map_txt_to_pb(kf_txt, mp_txt))
    assert (k2, m2) == (kf_txt, mp_txt)
    # empty map, and a keyframe without keypoints / connections (empty lines)
    assert _lib.map_pb_to_txt(b"\x0a\x00\x12\x00") == ("0 \n", "")
    e = MD()
    e.keyframes.next_id = 3
    e.keyframes.scale_factors.extend([1.0, 1.2])
    k = e.keyframes.keyframes.add()
    k.id = 2
    k.pose.rotation.extend([1, 0, 0, 0, 1, 0, 0, 0, 1])
    k.pose.translation.extend([0.5, 0, -2])
    e.mappoints.SetInParent()
    kt, mt = _lib.map_pb_to_txt(e.SerializeToString())
    assert kt == "3 1 1.2 \n2 0 0 0 0\n\n\n\n\n1 0 0 0 1 0 0 0 1 0.5 0 -2 \n\n\n\n\n" and mt == ""
    eb = MD()
    eb.ParseFromString(_lib.map_txt_to_pb(kt, mt))
    assert eb.keyframes.keyframes[0].id == 2 and list(eb.keyframes.keyframes[0].pose.translation) == [0.5, 0.0, -2.0]
