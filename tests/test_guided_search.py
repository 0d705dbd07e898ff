"""Grid-guided matching (VirtualFrame::findFeaturesInArea + getBestMatch, the core of ORBMatcher::searchByProjection):
oracle known answers (CPU) and the device kernels against the oracle (GPU)."""
import numpy as np
import pytest

from orb_slam2_ros2_amd import synth


def _queries(kps, desc, rng, n, th=15.0, scale=1.2):
    """frame-to-frame style queries (ORBMatcher.cc:288-314): search around a feature of the 'last' frame, octave window by motion"""
    pick = rng.integers(0, len(kps), n)
    qxy = np.stack([kps["x"][pick] + rng.normal(0, 3, n), kps["y"][pick] + rng.normal(0, 3, n)], 1).astype(np.float32)
    octv = kps["octave"][pick]
    sf2 = (np.float32(scale) ** octv.astype(np.float32)) ** 2          # getScaledFactor2
    radius = (np.float32(th) * sf2).astype(np.float32)
    mode = rng.integers(0, 3, n)
    lo = np.where(mode == 0, octv, np.where(mode == 1, 0, np.maximum(0, octv - 1))).astype(np.int8)
    hi = np.where(mode == 0, 7, np.where(mode == 1, octv, np.minimum(octv + 1, 7))).astype(np.int8)
    qd = desc[pick].copy()
    flips = rng.integers(0, 256, (n, 12))
    for i in range(n):
        for b in flips[i]:
            qd[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return qxy, radius, lo, hi, qd, pick


def test_oracle_candidate_sets_and_order(orc, kitti_pair):
    L, _ = kitti_pair
    kps, desc = orc.extractor(L).extract()
    rng = np.random.default_rng(5)
    qxy, radius, lo, hi, qd, pick = _queries(kps, desc, rng, 200)
    bi, bd, sd, nc = orc.search_in_area(kps, desc, 1241, 376, qxy, radius, lo, hi, qd)
    rows, cols = -(-376 // 48), -(-1241 // 64)
    for q in range(0, 200, 7):
        x, y, r = qxy[q, 0], qxy[q, 1], radius[q]
        x0, x1 = max(0, int(np.rint(np.float32(x - r)))), min(1241, int(np.rint(np.float32(x + r))))
        y0, y1 = max(0, int(np.rint(np.float32(y - r)))), min(376, int(np.rint(np.float32(y + r))))
        c0, c1, r0, r1 = min(cols - 1, x0 // 64), min(cols - 1, x1 // 64), min(rows - 1, y0 // 48), min(rows - 1, y1 // 48)
        cell_r = np.minimum(rows - 1, np.floor(kps["y"] / np.float32(48)).astype(int))
        cell_c = np.minimum(cols - 1, np.floor(kps["x"] / np.float32(64)).astype(int))
        cand = [i for rr in range(r0, r1 + 1) for cc in range(c0, c1 + 1)
                for i in np.nonzero((cell_r == rr) & (cell_c == cc) & (kps["octave"] >= lo[q]) & (kps["octave"] <= hi[q]))[0]]
        assert nc[q] == len(cand)
        if cand:
            assert (bi[q], bd[q], sd[q]) == orc.best_match(qd[q], desc, np.asarray(cand))[:3]
        else:
            assert bi[q] == -1
    hit = bi == pick
    assert hit.mean() > 0.5                      # most perturbed copies find their source feature
    # exclusion mask: the excluded feature can no longer win
    ex = np.zeros(len(kps), np.uint8)
    ex[pick[hit]] = 1
    bi2, _, _, nc2 = orc.search_in_area(kps, desc, 1241, 376, qxy, radius, lo, hi, qd, ex)
    assert not np.any(bi2[hit] == pick[hit]) and np.all(nc2 <= nc)


@pytest.mark.gpu
# (8000 features: the cells' lists no longer fit the grid kernel's LDS and are filled and sorted in device memory)
@pytest.mark.parametrize("w,h,nf", [(1241, 376, 2000), (640, 480, 1000), (900, 300, 3000), (1241, 376, 8000)])
def test_device_guided_search_matches_oracle(orc, w, h, nf):
    from orb_slam2_ros2_amd import ORBMatcher
    from orb_slam2_ros2_amd._lib import Context
    img, _ = synth.stereo_pair(50, w, h, n_rect=200)
    ctx = Context(w, h, n_features=nf, max_images=2)
    filler, _ = synth.stereo_pair(51, w, h, n_rect=200)
    (_, _), (kps, desc) = ctx.extract_batch([filler, img])            # search slot 1, not the default slot
    rng = np.random.default_rng(6)
    qxy, radius, lo, hi, qd, pick = _queries(kps, desc, rng, 700, th=15.0)
    qxy[:5] = [[0, 0], [w, h], [w - 1, 1], [w * 2, h * 2], [-50, 10]]     # boxes touching / outside the image
    radius[5:10] = [0.0, 0.4, 2000.0, 64.0, 48.0]
    ex = (rng.integers(0, 4, len(kps)) == 0).astype(np.uint8)
    for exclude in (None, ex):
        got = ORBMatcher.searchInArea(ctx, 1, qxy, radius, lo, hi, qd, exclude)
        ref = orc.search_in_area(kps, desc, w, h, qxy, radius, lo, hi, qd, exclude)
        for g, r, name in zip(got, ref, ("best_idx", "best_dist", "second_dist", "n_cand")):
            assert np.array_equal(g, r), name
    assert got[3].max() > 128      # some query exercises the multi-chunk path (more than two 64-candidate chunks)
    ctx.close()


@pytest.mark.gpu
def test_kept_grid_follows_the_slots_keypoints(orc):
    """the grid of a slot is kept between searches (Tracking searches a frame two to four times): a new extraction into the slot, an
    in-place undistortion (orbfe_frame_rgbd) and other frame bounds must each rebuild it"""
    from orb_slam2_ros2_amd import ORBMatcher
    from orb_slam2_ros2_amd._lib import Context
    w, h, nf = 640, 480, 1000
    ctx = Context(w, h, n_features=nf, max_images=2)
    rng = np.random.default_rng(3)
    tum = dict(fx=520.908620, fy=521.007327, cx=325.141442, cy=249.701764, k1=0.231222, k2=-0.784899, p1=-0.003257, p2=-0.000105, k3=0.917205, bf=40.0)
    for f in (60, 61):
        img = synth.mono_image(f, w, h)
        kps, desc = ctx.extract_slot(1, img)
        qxy, radius, lo, hi, qd, _ = _queries(kps, desc, rng, 400, th=15.0)
        ref = orc.search_in_area(kps, desc, w, h, qxy, radius, lo, hi, qd, None)
        for rep in range(3):                                   # the second and third search use the kept grid
            got = ORBMatcher.searchInArea(ctx, 1, qxy, radius, lo, hi, qd, None)
            assert all(np.array_equal(g, r) for g, r in zip(got, ref)), (f, rep)
        ku, _, _ = ctx.frame_rgbd(1, tum)                      # Camera::undistortPoints in place: the keypoints move
        assert (ku["x"][:len(kps)] != kps["x"]).sum() > len(kps) // 2
        ref_u = orc.search_in_area(ku[:len(kps)], desc, w, h, qxy, radius, lo, hi, qd, None)
        got_u = ORBMatcher.searchInArea(ctx, 1, qxy, radius, lo, hi, qd, None)
        assert all(np.array_equal(g, r) for g, r in zip(got_u, ref_u)), f
        assert not all(np.array_equal(g, r) for g, r in zip(got_u, ref))      # ... and the answers with them
    ctx.close()
