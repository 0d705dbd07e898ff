"""GPU tests of orbfe_frame_stereo / orbfe_frame_stereo_slots (Frame::createStereo's device work as one call, include/ORB_SLAM2/Frame.h:313-323):
bit-exact against the oracle and the golden digests, equal to the two calls it stands for (orbfe_extract_batch + orbfe_stereo_match), stable under
graph replay with changing images, usable beside the slot calls and followed by a plain orbfe_stereo_match on the same slots."""
import hashlib
import json
import os
import threading

import numpy as np
import pytest

from orb_slam2_ros2_amd import synth

pytestmark = pytest.mark.gpu

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.json")))
FX, BF = 718.856, 718.856 * 0.537166


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def lib():
    from orb_slam2_ros2_amd import _lib
    return _lib


def assert_same(a, b):
    (alk, ald), (ark, ard), anm, aru, adp = a
    (blk, bld), (brk, brd), bnm, bru, bdp = b
    assert alk.tobytes() == blk.tobytes() and ark.tobytes() == brk.tobytes(), "keypoints"
    assert np.array_equal(ald, bld) and np.array_equal(ard, brd), "descriptors"
    assert anm == bnm, f"n_matches {anm} vs {bnm}"
    assert np.array_equal(aru.view(np.int64), bru.view(np.int64)) and np.array_equal(adp.view(np.int64), bdp.view(np.int64)), "right_u / depth"


def two_calls(ctx, L, R):
    l, r = ctx.extract_batch([L, R])
    nm, ru, dp, _, _ = ctx.stereo_match(0, 1, FX, BF)
    return l, r, nm, ru, dp


@pytest.mark.parametrize("f", [3, 11])
def test_row_table_by_one_workgroup_equals_the_eight_part_table(lib, monkeypatch, f):
    """The right image's stereo row table rides in the descriptor launch of a frame or two: by the launch's eight spare workgroups (r6,
    rowtable_build_part: an eighth of the rows each, totals through flags) or, with ORBFE_ROWTABLE_ONE_WG=1, by one (r5).  Same matches, bit
    for bit, through both call shapes; a context is built per setting (the switch is read at orbfe_create)."""
    L, R = synth.stereo_pair(f)
    res = []
    for one in (False, True):
        if one:
            monkeypatch.setenv("ORBFE_ROWTABLE_ONE_WG", "1")
        ctx = lib.Context(1241, 376, max_images=2)
        try:
            a = two_calls(ctx, L, R)
            b = ctx.frame_stereo(L, R, FX, BF)
            assert_same(a, b)
            res.append(a)
        finally:
            ctx.close()
    assert_same(res[0], res[1])


@pytest.mark.parametrize("f", [0, 1, 7])
def test_one_call_frame_against_oracle_and_golden(orc, lib, f):
    L, R = synth.stereo_pair(f)
    ctx = lib.Context(1241, 376, max_images=2)
    try:
        for rep in range(3):  # the first call captures the launch sequence, the others replay it
            (lk, ld), (rk, rd), nm, ru, dp = ctx.frame_stereo(L, R, FX, BF)
            g = G["frames"][f"kitti_{f}"]
            n = len(lk)
            assert (sha(lk), sha(ld), sha(rk), sha(rd)) == (g["lk_sha"], g["ld_sha"], g["rk_sha"], g["rd_sha"]), f"rep {rep}"
            assert sha(ru[:n]) == g["right_u_sha"] and sha(dp[:n]) == g["depth_sha"] and nm == g["n_matches"], f"rep {rep}"
            assert np.all(ru[n:] == -1.0) and np.all(dp[n:] == -1.0)
        exl, exr = orc.extractor(L), orc.extractor(R)
        (okl, odl), (okr, odr) = exl.extract(), exr.extract()
        om, oru, odp, _, _ = exl.stereo_match(exr, okl, odl, okr, odr, FX, BF)
        assert nm == om and np.array_equal(ru[:n].view(np.int64), oru.view(np.int64)) and np.array_equal(dp[:n].view(np.int64), odp.view(np.int64))
        assert lk.tobytes() == okl.tobytes() and np.array_equal(ld, odl) and rk.tobytes() == okr.tobytes() and np.array_equal(rd, odr)
    finally:
        ctx.close()


def test_equals_the_two_calls_with_changing_frames_and_interleaved_entry_points(lib):
    ctx, ref = lib.Context(1241, 376, max_images=2), lib.Context(1241, 376, max_images=2)
    try:
        for f in (3, 4, 3, 9, 0):
            L, R = synth.stereo_pair(f)
            assert_same(ctx.frame_stereo(L, R, FX, BF), two_calls(ref, L, R))
            # a plain match on the same slots afterwards: the pair's counter has been counted into and must be cleared first
            nm, ru, dp, _, _ = ctx.stereo_match(0, 1, FX, BF)
            nm2, ru2, dp2, _, _ = ref.stereo_match(0, 1, FX, BF)
            assert nm == nm2 and np.array_equal(ru, ru2) and np.array_equal(dp, dp2)
            # and the two-call path on the SAME context between fused calls (separate graphs, shared slots)
            if f == 4:
                assert_same(two_calls(ctx, L, R), two_calls(ref, L, R))
        # other camera constants: a different captured sequence, not a stale one
        L, R = synth.stereo_pair(5)
        a = ctx.frame_stereo(L, R, 500.0, 200.0)
        l, r = ref.extract_batch([L, R])
        nm, ru, dp, _, _ = ref.stereo_match(0, 1, 500.0, 200.0)
        assert_same(a, (l, r, nm, ru, dp))
    finally:
        ctx.close(), ref.close()


def test_padded_rows_and_other_geometry(lib):
    w, h = 640, 480
    ctx, ref = lib.Context(w, h, n_features=1000, max_images=2), lib.Context(w, h, n_features=1000, max_images=2)
    try:
        L, R = synth.stereo_pair(2, w, h)
        pad = np.zeros((2, h, w + 24), np.uint8)
        pad[0, :, :w], pad[1, :, :w] = L, R
        assert_same(ctx.frame_stereo(pad[0, :, :w], pad[1, :, :w], 520.0, 40.0), (*ref.extract_batch([L, R]), *ref.stereo_match(0, 1, 520.0, 40.0)[:3]))
    finally:
        ctx.close(), ref.close()


def test_slot_pairs_beside_slot_calls_on_other_threads(lib):
    ctx, ref = lib.Context(1241, 376, max_images=6), lib.Context(1241, 376, max_images=2)
    try:
        frames = [synth.stereo_pair(f) for f in (0, 1, 2)]
        want = [two_calls(ref, L, R) for L, R in frames]
        for rep in range(2):
            for k, (L, R) in enumerate(frames):
                assert_same(ctx.frame_stereo(L, R, FX, BF, slot_left=2 * k), want[k])
        # the device-side results of an older pair are still there (its slots have not been written since)
        nm, ru, dp, _, _ = ctx.stereo_match(0, 1, FX, BF)
        assert nm == want[0][2] and np.array_equal(ru, want[0][3])
        # two threads, two pairs, twenty frames each; a third thread extracts single images into the last pair's slots
        errs = []

        def worker(k):
            try:
                for it in range(20):
                    assert_same(ctx.frame_stereo(*frames[k], FX, BF, slot_left=2 * k), want[k])
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        def single():
            try:
                for it in range(20):
                    kp, d = ctx.extract_slot(4 + (it & 1), frames[2][it & 1])
                    assert kp.tobytes() == want[2][it & 1][0].tobytes() and np.array_equal(d, want[2][it & 1][1])
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        th = [threading.Thread(target=worker, args=(0,)), threading.Thread(target=worker, args=(1,)), threading.Thread(target=single)]
        [t.start() for t in th], [t.join() for t in th]
        assert not errs, errs
    finally:
        ctx.close(), ref.close()


def test_context_without_per_slot_row_tables(lib):
    """more than sixteen slots: the extraction builds no per-slot row table, the match inside the call builds the pair's own"""
    ctx, ref = lib.Context(1241, 376, max_images=32), lib.Context(1241, 376, max_images=2)
    try:
        for f in (6, 2, 6):
            L, R = synth.stereo_pair(f)
            want = two_calls(ref, L, R)
            assert_same(ctx.frame_stereo(L, R, FX, BF), want)
            assert_same(ctx.frame_stereo(L, R, FX, BF, slot_left=30), want)
    finally:
        ctx.close(), ref.close()


def test_argument_checks(lib):
    L, R = synth.stereo_pair(0)
    one = lib.Context(1241, 376, max_images=1)
    ctx = lib.Context(1241, 376, max_images=4)
    try:
        with pytest.raises(lib.OrbfeError):
            one.frame_stereo(L, R, FX, BF)
        for bad in (1, 3, 4, -2):
            with pytest.raises(lib.OrbfeError):
                ctx.frame_stereo(L, R, FX, BF, slot_left=bad)
        with pytest.raises(ValueError):
            ctx.frame_stereo(L[:100], R, FX, BF)
        ctx.frame_stereo(L, R, FX, BF, slot_left=2)  # still usable
    finally:
        one.close(), ctx.close()


def test_extract_slot_in_two_halves_equals_the_one_call(orc):
    """orbfe_extract_slot_begin / _end (ABI 4): the constructor-time start of the reference's extractor (Frame.cc:91-92 builds both
    ORBExtractor objects before the extract() threads exist) -- same results as orbfe_extract_slot, the halves on different threads, both
    eyes outstanding at once, the slot refusing other calls in between, a drain without outputs."""
    import threading
    from orb_slam2_ros2_amd._lib import Context, OrbfeError
    L, R = synth.stereo_pair(2)
    ctx = Context(1241, 376, max_images=4)
    want_l, want_r = ctx.extract_slot(0, L), ctx.extract_slot(1, R)
    ctx.extract_slot_begin(2, L)
    ctx.extract_slot_begin(3, R)                 # both outstanding
    with pytest.raises(OrbfeError):
        ctx.extract_slot(2, L)                   # the slot is busy until _end
    with pytest.raises(OrbfeError):
        ctx.extract_slot_begin(2, L)
    # ... and so does every other entry point that names the slot (they run on the context stream, which does not wait for the lane)
    with pytest.raises(OrbfeError, match="outstanding"):
        ctx.pyramid(2, 0)
    with pytest.raises(OrbfeError, match="outstanding"):
        ctx.stereo_match(2, 3, FX, BF)
    with pytest.raises(OrbfeError, match="outstanding"):
        ctx.stereo_match(0, 3, FX, BF)           # the right slot alone is pending
    with pytest.raises(OrbfeError, match="outstanding"):
        ctx.fetch_features(3)
    with pytest.raises(OrbfeError, match="outstanding"):
        ctx.extract_batch([L, R, L])             # would write slots 0..2
    assert np.array_equal(ctx.extract_batch([L, R])[0][0], want_l[0])   # slots 0, 1 are idle: served, and the begun slots are untouched
    got = {}
    th = [threading.Thread(target=lambda s=s: got.__setitem__(s, ctx.extract_slot_end(s))) for s in (2, 3)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    for (k, d), (wk, wd) in ((got[2], want_l), (got[3], want_r)):
        assert np.array_equal(k, wk) and np.array_equal(d, wd)
    with pytest.raises(OrbfeError):
        ctx.extract_slot_end(2)                  # nothing outstanding
    nm, ru, dp, _, _ = ctx.stereo_match(2, 3, FX, BF)      # the results are resident in the slots as after orbfe_extract_slot
    ref = orc.stereo_frame(L, R, fx=FX, bf=BF)
    assert nm == ref["n_matches"] and np.array_equal(ru[:len(want_l[0])], ref["right_u"])
    for _ in range(3):                           # repeated use of one slot (the captured launch sequence is replayed)
        ctx.extract_slot_begin(0, R)
        k, d = ctx.extract_slot_end(0)
        assert np.array_equal(k, want_r[0]) and np.array_equal(d, want_r[1])
    ctx.extract_slot_begin(1, L)
    ctx.lib.orbfe_extract_slot_end(ctx.h, 1, None, None, None)   # drain: no outputs
    assert np.array_equal(ctx.extract_slot(1, L)[0], want_l[0])
    ctx.close()
