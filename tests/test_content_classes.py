"""The content classes of synth.stereo_pair_content: CPU -- the generator and the oracle against
tests/golden/golden_v4.json (tools/make_golden_v4.py); GPU -- the HIP path against the oracle on every class, arrays and digests,
through the single-frame calls and through the batched call bench.py's content sweep times.

What the classes are for (ORBExtractor.cc:346-375): "camera" sends more than a third of the cells through the second cv::FAST call at
the low threshold, "saturated" hands the quadtree several times the candidates of the class the headline is quoted on, "sparse"
leaves the fine levels below their quota (quirk Q3: they return nothing)."""
import hashlib
import json
import os

import numpy as np
import pytest

from orb_slam2_ros2_amd import synth
from orb_slam2_ros2_amd.digest import batch_digests, pair_digest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G4 = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_v4.json")))["classes"]
FX, BF = 718.856, 718.856 * 0.537166
CLASSES = [c for c in synth.CONTENT_CLASSES if c != "rect"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_rect_class_is_the_headline_generator():
    a, b = synth.stereo_pair_content(3, "rect", 320, 200), synth.stereo_pair(3, 320, 200)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    with pytest.raises(ValueError):
        synth.stereo_pair_content(0, "fog")


@pytest.mark.parametrize("cls", CLASSES)
def test_oracle_on_content_class_matches_golden_v4(orc, cls):
    g = G4[cls]["frames"]["0"]
    L, R = synth.stereo_pair_content(0, cls)
    assert L.dtype == np.uint8 and L.shape == (376, 1241) and sha(L) == g["left_sha"] and sha(R) == g["right_sha"]
    r = orc.stereo_frame(L, R, fx=FX, bf=BF, math_mode=0, threads=2)
    assert (len(r["lk"]), len(r["rk"]), int(r["n_matches"])) == (g["n_left"], g["n_right"], g["n_matches"])
    assert pair_digest(r["lk"], r["ld"], r["rk"], r["rd"], r["right_u"], r["depth"], r["n_matches"]) == G4[cls]["pairs"]["0"]
    ex = orc.extractor(L)
    k, _ = ex.extract()
    lo = cells = 0
    for l in range(8):
        c = ex.candidates(l)
        assert len(c) == g["candidates_per_level"][l]
        wl, hl = ex.level_info(l)[:2]
        a, b = synth.lo_pass_cells(c, wl - 32, hl - 32)
        lo, cells = lo + a, cells + b
    assert [int((k["octave"] == l).sum()) for l in range(8)] == g["selected_per_level"]
    assert (lo, cells) == (g["cells_lo_pass"], g["cells"])


def test_the_classes_span_what_they_are_meant_to():
    """the properties the classes exist for, from the committed oracle statistics (frames 0 and 1)"""
    for f in ("0", "1"):
        cam, sat, spa = (G4[c]["frames"][f] for c in ("camera", "saturated", "sparse"))
        assert cam["cells_lo_pass"] > 0.3 * cam["cells"] and cam["n_left"] == 2000        # a third of the cells repeat at 7, still 2000 features
        assert sat["cells_lo_pass"] == 0 and sum(sat["candidates_per_level"]) > 35000     # corners everywhere, ~19 x the quota
        assert min(sat["candidates_per_level"][l] for l in range(8)) > 1000
        assert sum(1 for n in spa["selected_per_level"] if n == 0) >= 3 and spa["n_left"] > 0   # Q3 on several levels, not on all


def test_lo_pass_cells_against_a_direct_count(orc):
    """the host-side statistic against its definition: a cell takes the low pass iff cv::FAST at the high threshold finds nothing in it"""
    L, _ = synth.stereo_pair_content(1, "camera")
    ex = orc.extractor(L)
    ex.extract()
    for l in (0, 5):
        plane = ex.plane(l)
        wl, hl = ex.level_info(l)[:2]
        w, h = wl - 32, hl - 32
        n_cols, n_rows = w // 30, h // 30
        wc, hc = w // n_cols, h // n_rows
        lo = cells = 0
        for i in range(n_rows):
            y0 = 16 + i * hc
            y1 = min(y0 + hc + 6, 16 + h)
            if y0 >= 16 + h - 6:
                continue
            for j in range(n_cols):
                x0 = 16 + j * wc
                x1 = min(x0 + wc + 6, 16 + w)
                if x0 >= 16 + w - 6:
                    continue
                cells += 1
                lo += len(orc.fast(np.ascontiguousarray(plane[y0:y1, x0:x1]), 20, True)) == 0
        assert synth.lo_pass_cells(ex.candidates(l), w, h) == (lo, cells)


# ---- GPU ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("cls", CLASSES)
def test_gpu_single_frames_of_content_class_equal_the_oracle(orc, cls):
    from orb_slam2_ros2_amd._lib import Context
    ctx = Context(1241, 376, max_images=2)
    for f in (0, 1, 2):
        L, R = synth.stereo_pair_content(f, cls)
        ref = orc.stereo_frame(L, R, fx=FX, bf=BF)
        (lk, ld), (rk, rd) = ctx.extract_batch([L, R])
        nm, ru, dp, _, _ = ctx.stereo_match(0, 1, FX, BF)
        n = len(lk)
        assert np.array_equal(lk, ref["lk"]) and np.array_equal(ld, ref["ld"]), f"{cls} frame {f}: left features"
        assert np.array_equal(rk, ref["rk"]) and np.array_equal(rd, ref["rd"]), f"{cls} frame {f}: right features"
        assert nm == ref["n_matches"] and np.array_equal(ru[:n].view(np.int64), ref["right_u"].view(np.int64)), f"{cls} frame {f}: right_u"
        assert np.array_equal(dp[:n].view(np.int64), ref["depth"].view(np.int64)), f"{cls} frame {f}: depth"
        assert pair_digest(lk, ld, rk, rd, ru, dp, nm) == G4[cls]["pairs"][str(f)]
        if f == 0:   # the candidate SETS per level (the quadtree's input), and the one-call frame
            ex = orc.extractor(L)
            ex.extract()
            for l in range(8):
                assert np.array_equal(ctx.debug_candidates(0, l), ex.candidates(l)), f"{cls}: level {l} candidates"
            (flk, fld), (frk, frd), fnm, fru, fdp = ctx.frame_stereo(L, R, FX, BF)
            assert pair_digest(flk, fld, frk, frd, fru, fdp, fnm) == G4[cls]["pairs"]["0"]
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cls", CLASSES)
def test_gpu_batch_of_content_class_equals_the_golden_digests(cls):
    """the batched device-resident call at a size whose launches run the cell loop of k_fast and the global-record quadtree (what the
    content sweep of bench.py times): 16 distinct frames x 16"""
    import torch
    from orb_slam2_ros2_amd._lib import Context
    B, U = 256, 16
    fr = [synth.stereo_pair_content(f, cls) for f in range(U)]
    ctx = Context(1241, 376, max_images=2 * B)
    dl = torch.from_numpy(np.stack([fr[i % U][0] for i in range(B)])).cuda()
    dr = torch.from_numpy(np.stack([fr[i % U][1] for i in range(B)])).cuda()
    for _ in range(2):
        ctx.stereo_batch_device(dl.data_ptr(), dr.data_ptr(), 1241, 1241 * 376, B, FX, BF)
    ctx.sync()
    kps, desc, cnt = ctx.fetch_batch(0, 2 * B)
    ru, dp, nm = ctx.fetch_stereo_batch(0, B)
    dig = batch_digests(kps, desc, cnt, ru, dp, nm)
    bad = [p for p in range(B) if dig[p] != G4[cls]["pairs"][str(p % U)]]
    assert not bad, f"{cls}: pairs {bad[:8]} differ from golden_v4"
    ctx.close()


@pytest.mark.gpu
def test_gpu_row_parallel_matcher_with_crowded_rows(orc):
    """k_stereo_rows takes a row's left keypoints in groups of 64 and its right candidates in chunks of 64: an image whose texture is one
    horizontal band puts all 2000 keypoints into ~60 rows (more than 64 left keypoints in a row, candidate lists of several chunks) --
    the batch call against the oracle, and against the one-wave-per-keypoint kernel the single-pair call uses."""
    import torch
    from orb_slam2_ros2_amd._lib import Context
    sat = [synth.stereo_pair_content(f, "saturated") for f in range(2)]
    pairs = []
    for L, R in sat:
        bl, br = np.full_like(L, 96), np.full_like(R, 96)
        bl[160:215] = L[160:215]
        br[160:215] = R[160:215]
        pairs.append((bl, br))
    ref = [orc.stereo_frame(L, R, fx=FX, bf=BF) for L, R in pairs]
    rows = np.round(ref[0]["lk"]["y"]).astype(int)
    assert np.bincount(rows).max() > 64 and len(ref[0]["lk"]) > 1000, "the premise: a row with more than 64 left keypoints"
    B = 8
    ctx = Context(1241, 376, max_images=2 * B)
    dl = torch.from_numpy(np.stack([pairs[i % 2][0] for i in range(B)])).cuda()
    dr = torch.from_numpy(np.stack([pairs[i % 2][1] for i in range(B)])).cuda()
    ctx.stereo_batch_device(dl.data_ptr(), dr.data_ptr(), 1241, 1241 * 376, B, FX, BF)
    ctx.sync()
    kps, desc, cnt = ctx.fetch_batch(0, 2 * B)
    ru, dp, nm = ctx.fetch_stereo_batch(0, B)
    for p in range(B):
        r = ref[p % 2]
        nl = len(r["lk"])
        assert cnt[2 * p] == nl and nm[p] == r["n_matches"], f"pair {p}: counts"
        assert np.array_equal(kps[2 * p, :nl], r["lk"]) and np.array_equal(desc[2 * p, :nl], r["ld"])
        assert np.array_equal(ru[p, :nl].view(np.int64), r["right_u"].view(np.int64)), f"pair {p}: right_u"
        assert np.array_equal(dp[p, :nl].view(np.int64), r["depth"].view(np.int64)), f"pair {p}: depth"
        assert (ru[p, nl:] == -1).all() and (dp[p, nl:] == -1).all()      # slots past the count keep the defaults
    ctx.close()
    c1 = Context(1241, 376, max_images=2)
    c1.extract_batch(list(pairs[0]))
    nm1, ru1, dp1, br1, bd1 = c1.stereo_match(0, 1, FX, BF)
    nl = len(ref[0]["lk"])
    assert nm1 == ref[0]["n_matches"] and np.array_equal(ru1[:nl].view(np.int64), ref[0]["right_u"].view(np.int64))
    c1.close()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,nf,nl,cls", [(752, 480, 1200, 8, "camera"), (320, 240, 500, 5, "saturated"), (1920, 1080, 3000, 8, "rect"),
                                           (1241, 376, 2000, 10, "camera")])   # ten levels: octaves 8, 9 in the matcher's packed record
def test_gpu_batch_matcher_at_other_geometries(orc, w, h, nf, nl, cls):
    """the batch path (k_rowtable + k_stereo_rows + k_stereo_sad) away from the KITTI shape: other row counts, quotas, level counts and
    focal lengths -- five pairs per call (the batch matcher starts at four), every pair against the oracle"""
    import torch
    from orb_slam2_ros2_amd._lib import Context
    fx, bf = 0.58 * w, 0.58 * w * 0.11
    B = 5
    pairs = [synth.stereo_pair_content(10 + f, cls, w, h) for f in range(B)]
    ref = [orc.stereo_frame(L, R, n_features=nf, n_levels=nl, fx=fx, bf=bf) for L, R in pairs]
    ctx = Context(w, h, n_features=nf, n_levels=nl, max_images=2 * B)
    dl = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda()
    dr = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    for _ in range(2):
        ctx.stereo_batch_device(dl.data_ptr(), dr.data_ptr(), w, w * h, B, fx, bf)
    ctx.sync()
    kps, desc, cnt = ctx.fetch_batch(0, 2 * B)
    ru, dp, nm = ctx.fetch_stereo_batch(0, B)
    assert sum(r["n_matches"] for r in ref) > 0
    if nl > 8:   # matches whose LEFT keypoint sits on a level the 3-bit octave field of r5 could not hold
        assert sum(int(((r["lk"]["octave"] >= 8) & (r["right_u"] >= 0)).sum()) for r in ref) > 0
    for p in range(B):
        r = ref[p]
        nl_, nr_ = len(r["lk"]), len(r["rk"])
        assert (cnt[2 * p], cnt[2 * p + 1], nm[p]) == (nl_, nr_, r["n_matches"]), f"pair {p}: counts"
        assert np.array_equal(kps[2 * p, :nl_], r["lk"]) and np.array_equal(desc[2 * p + 1, :nr_], r["rd"])
        assert np.array_equal(ru[p, :nl_].view(np.int64), r["right_u"].view(np.int64)), f"pair {p}: right_u"
        assert np.array_equal(dp[p, :nl_].view(np.int64), r["depth"].view(np.int64)), f"pair {p}: depth"
    ctx.close()
