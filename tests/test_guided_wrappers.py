"""Host logic of the guided searches (ORBMatcher::searchByProjection x2, searchByBow, verifyAngle; ORBMatcher.cc:170-347, 561-612,
1013-1051) in the Python mirror.  CPU: the batched formulation with the oracle injected for the device calls, against loops written
line by line after the reference (one query at a time, getBestMatch over an explicit candidate list).  GPU: the same through the
device calls."""
import numpy as np
import pytest

from orb_slam2_ros2_amd import ORBMatcher

W, H = 1241, 376
SF = (np.float32(1.2) ** np.arange(8, dtype=np.float32)).astype(np.float32)


def _cands(kps, x, y, r, lo, hi):
    """VirtualFrame::findFeaturesInArea (Frame.cc:286-311): features of the overlapping 64x48 cells, rows outer, columns inner"""
    rows, cols = -(-H // 48), -(-W // 64)
    x0, x1 = max(0, int(np.rint(np.float32(x - r)))), min(W, int(np.rint(np.float32(x + r))))
    y0, y1 = max(0, int(np.rint(np.float32(y - r)))), min(H, int(np.rint(np.float32(y + r))))
    c0, c1, r0, r1 = min(cols - 1, x0 // 64), min(cols - 1, x1 // 64), min(rows - 1, y0 // 48), min(rows - 1, y1 // 48)
    cr = np.minimum(rows - 1, np.floor(kps["y"] / np.float32(48)).astype(int))
    cc = np.minimum(cols - 1, np.floor(kps["x"] / np.float32(64)).astype(int))
    return [int(i) for rr in range(r0, r1 + 1) for c in range(c0, c1 + 1)
            for i in np.nonzero((cr == rr) & (cc == c) & (kps["octave"] >= lo) & (kps["octave"] <= hi))[0]]


def _best(orc, q, desc, cand):
    bi, bd, sd = orc.best_match(q, desc, np.asarray(cand))[:3]
    return int(bi), int(bd), np.float32(bd) / np.float32(sd)


@pytest.fixture(scope="module")
def frames(orc, kitti_pair):
    L, R = kitti_pair
    k1, d1 = orc.extractor(L).extract()
    k2, d2 = orc.extractor(R).extract()          # "last frame": the right image, shifted content
    return k1, d1, k2, d2


def _csr_match(orc):
    """the CSR form of orbfe_match_bruteforce, on the oracle's single-query getBestMatch"""
    def run(q, t, off, cand):
        res = [orc.best_match(q[i], t, cand[off[i]:off[i + 1]])[:3] for i in range(len(q))]
        return tuple(np.array([r[k] for r in res], np.int32) for k in range(3))
    return run


def _inject(orc, k1, d1):
    return lambda *a: orc.search_in_area(k1, d1, W, H, *a)


@pytest.mark.parametrize("z,bFuse", [(0.0, False), (0.9, False), (-0.9, False), (0.0, True)])
def test_search_by_projection_frames(orc, frames, z, bFuse):
    k1, d1, k2, d2 = frames
    rng = np.random.default_rng(3)
    valid2 = rng.random(len(k2)) < 0.7
    has1 = rng.random(len(k1)) < 0.3
    inv = rng.random(len(k2)) < 0.8
    m = ORBMatcher(0.7)
    got = m.searchByProjectionFrames(None, 0, SF, k2, d2, valid2, has1, 15.0, z, 0.54, bFuse, inv if bFuse else None,
                                     area_search=_inject(orc, k1, d1))
    want = []
    up, down = (abs(z) > 0.54 and z > 0), (abs(z) > 0.54 and z <= 0)
    for idx in range(len(k2)):
        if not valid2[idx] or (bFuse and not inv[idx]):
            continue
        o = int(k2["octave"][idx])
        lo, hi = (o, 7) if up else ((0, o) if down else (max(0, o - 1), min(o + 1, 7)))
        cand = _cands(k1, k2["x"][idx], k2["y"][idx], np.float32(15.0) * SF[o] * SF[o], lo, hi)
        if not bFuse:
            cand = [c for c in cand if not has1[c]]
        if not cand:
            continue
        bi, bd, ratio = _best(orc, d2[idx], d1, cand)
        if ratio < np.float32(0.7) and bd < 50:
            want.append((bi, idx, bd))
    assert got == want and len(got) > 5


def test_search_by_projection_map_points(orc, frames):
    k1, d1, k2, d2 = frames
    rng = np.random.default_rng(4)
    n = len(k2)
    uv = np.stack([k2["x"], k2["y"]], 1) + rng.normal(0, 2, (n, 2)).astype(np.float32)
    level, cos = k2["octave"].astype(np.int32), rng.uniform(0.5, 1.0, n).astype(np.float32)
    usable = rng.random(n) < 0.8
    has = rng.random(len(k1)) < 0.25
    m = ORBMatcher(0.8)
    for bFuse in (False, True):
        got, n_got = m.searchByProjectionMapPoints(None, 0, uv, level, cos, d2, usable, 1.0, has, bFuse, scale_factors=SF,
                                                   area_search=_inject(orc, k1, d1))
        want, n_want, cur = [], (0 if bFuse else int(has.sum())), has.copy()
        for idx in range(n):
            if not usable[idx]:
                continue
            rad = (np.float32(2.5) if cos[idx] > np.float32(0.998) else np.float32(4.0)) * np.float32(1.0) * SF[level[idx]] * SF[level[idx]]
            cand = _cands(k1, uv[idx, 0], uv[idx, 1], rad, max(0, level[idx] - 1), min(7, level[idx] + 1))
            if not cand:
                continue
            bi, bd, ratio = _best(orc, d2[idx], d1, cand)
            if bd < 50 and ratio < np.float32(0.8):
                if bFuse:
                    want.append((bi, idx, bd))
                    n_want += 1
                elif not cur[bi]:
                    cur[bi] = True
                    want.append((bi, idx))
                    n_want += 1
        assert got == want and n_got == n_want and len(got) > 5


@pytest.mark.parametrize("bAddMPs,bLoop", [(False, False), (True, False), (False, True)])
def test_search_by_bow(orc, frames, bAddMPs, bLoop):
    k1, d1, k2, d2 = frames
    rng = np.random.default_rng(6)
    # stand-in for the DBoW feature vectors: node id = a coarse hash of the position, so that true correspondences share a node
    node = lambda k: ((k["x"] // 160).astype(int) * 8 + (k["y"] // 94).astype(int) + 3 * k["octave"]).astype(int)
    fv1, fv2 = {}, {}
    for i, nd in enumerate(node(k1)):
        fv1.setdefault(int(nd), []).append(i)
    for i, nd in enumerate(node(k2)):
        fv2.setdefault(int(nd), []).append(i)
    g1, i1, g2, i2 = (rng.random(len(k)) < p for k, p in ((k1, 0.4), (k1, 0.7), (k2, 0.6), (k2, 0.7)))
    m = ORBMatcher(0.75, True)
    got = m.searchByBow(None, d1, d2, fv1, fv2, g1, i1, g2, i2, k1["angle"], k2["angle"], bAddMPs, bLoop, best_match=_csr_match(orc))
    raw = []
    for nd in sorted(set(fv1) & set(fv2)):
        for pk in fv2[nd]:
            good = g2[pk]
            if bAddMPs:
                if good and i2[pk]:
                    continue
            elif not bLoop and not good:
                continue
            cand = []
            for p in fv1[nd]:
                gf = g1[p]
                if bAddMPs:
                    if gf and i1[p]:
                        continue
                    cand.append(p)
                elif bLoop:
                    cand.append(p)
                elif not gf:
                    cand.append(p)
            if not cand:
                continue
            bi, bd, ratio = _best(orc, d2[pk], d1, cand)
            if bd > 50 or ratio > np.float32(0.75):
                continue
            raw.append((bi, pk, bd))
    want = ORBMatcher.verifyAngle(raw, k1["angle"], k2["angle"])
    assert got == want and len(raw) > 3


def test_verify_angle_known_answers():
    a1 = np.array([10, 10, 10, 200, 200, 50, 359.5, 0.0], np.float32)
    a2 = np.array([0, 1, 2, 100, 101, 300, 0.0, 0.5], np.float32)
    ms = [(i, i, 0) for i in range(8)]
    # differences: 10, 9, 8 -> bin 0 | 100, 99 -> bin 8 | -250 -> 110 -> bin 9 | 359.5 -> bin 29 | -0.5 -> 359.5 -> bin 29
    out = ORBMatcher.verifyAngle(ms, a1, a2)
    assert out == [(0, 0, 0), (1, 1, 0), (2, 2, 0), (3, 3, 0), (4, 4, 0), (6, 6, 0), (7, 7, 0)]     # bins 0, 8, 29; ordered by bin id
    assert ORBMatcher.verifyAngle([], a1, a2) == []
    assert ORBMatcher.verifyAngle(ms[:2], a1, a2) == ms[:2]                                        # fewer bins than mnBinChoose


@pytest.mark.gpu
def test_device_guided_wrappers_match_the_oracle_path(orc, kitti_pair):
    from orb_slam2_ros2_amd._lib import Context
    L, R = kitti_pair
    ctx = Context(W, H, max_images=2)
    (k1, d1), (k2, d2) = ctx.extract_batch([L, R])
    rng = np.random.default_rng(3)
    valid2, has1 = rng.random(len(k2)) < 0.7, rng.random(len(k1)) < 0.3
    m = ORBMatcher(0.7)
    for z, bFuse in ((0.0, False), (0.9, False), (-0.9, True)):
        dev = m.searchByProjectionFrames(ctx, 0, SF, k2, d2, valid2, has1, 15.0, z, 0.54, bFuse)
        ref = m.searchByProjectionFrames(None, 0, SF, k2, d2, valid2, has1, 15.0, z, 0.54, bFuse, area_search=_inject(orc, k1, d1))
        assert dev == ref and len(dev) > 5
    uv = np.stack([k2["x"], k2["y"]], 1)
    dev = m.searchByProjectionMapPoints(ctx, 0, uv, k2["octave"], np.full(len(k2), 0.9, np.float32), d2, valid2, 1.0, has1)
    ref = m.searchByProjectionMapPoints(None, 0, uv, k2["octave"], np.full(len(k2), 0.9, np.float32), d2, valid2, 1.0, has1, scale_factors=SF,
                                        area_search=_inject(orc, k1, d1))
    assert dev == ref and dev[1] > has1.sum()
    fv1 = {int(n): list(np.flatnonzero(k1["octave"] == n)) for n in range(8)}
    fv2 = {int(n): list(np.flatnonzero(k2["octave"] == n)) for n in range(8)}
    g = np.zeros(len(k1), bool), np.ones(len(k1), bool), np.ones(len(k2), bool), np.ones(len(k2), bool)
    dev = m.searchByBow(ctx, d1, d2, fv1, fv2, *g, k1["angle"], k2["angle"])
    ref = m.searchByBow(None, d1, d2, fv1, fv2, *g, k1["angle"], k2["angle"], best_match=_csr_match(orc))
    assert dev == ref and len(dev) > 5
    ctx.close()


# ---- MapPoint::isInVision / predictLevel (src/MapPoint.cc:141-201) ----------------------------------------------------------------
def _vision_case(n=4000, seed=5):
    rng = np.random.default_rng(seed)
    ang = 0.3
    Rcw = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
    tcw = np.array([0.2, -0.1, 0.4], np.float32)
    pos = np.stack([rng.uniform(-6, 6, n), rng.uniform(-3, 3, n), rng.uniform(-2, 12, n)], 1).astype(np.float32)  # some behind the camera
    vd = rng.normal(size=(n, 3)).astype(np.float32)
    vd[: n // 2] = (pos[: n // 2] * rng.uniform(0.5, 1.5, (n // 2, 1))).astype(np.float32)  # half of them roughly along the viewing ray
    ref = np.linalg.norm(pos, axis=1).astype(np.float32)
    max_d = (ref * rng.uniform(0.7, 3.0, n)).astype(np.float32)
    min_d = (ref * rng.uniform(0.1, 1.1, n)).astype(np.float32)
    cam = (520.9086, 521.0073, 325.1414, 249.7018)
    bounds = (0.0, 640.0, 0.0, 480.0)
    return pos, vd, max_d, min_d, Rcw, tcw, cam, bounds


def test_is_in_vision_restatement_against_definition():
    from oracle import pyoracle
    orc = pyoracle.Oracle(pyoracle.build())
    pos, vd, max_d, min_d, Rcw, tcw, cam, bounds = _vision_case()
    r = orc.project_map_points(pos, vd, max_d, min_d, Rcw, tcw, cam, bounds)
    # fp64 evaluation of the five conditions: the float restatement may only differ within rounding of a boundary
    pc = pos.astype(np.float64) @ Rcw.astype(np.float64).T + tcw
    d = np.linalg.norm(pc, axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        u, v = pc[:, 0] / pc[:, 2] * cam[0] + cam[2], pc[:, 1] / pc[:, 2] * cam[1] + cam[3]
        w = vd.astype(np.float64) @ Rcw.astype(np.float64).T
        cos = (w * pc).sum(1) / (d * np.linalg.norm(w, axis=1))
    vis = (pc[:, 2] >= 0) & (d < max_d) & (d > min_d) & (u > 0) & (u < 640) & (v > 0) & (v < 480) & (cos >= 0.5)
    margin = np.minimum.reduce([np.abs(d - max_d), np.abs(d - min_d), np.abs(u), np.abs(640 - u), np.abs(v), np.abs(480 - v),
                                np.abs(cos - 0.5) * 100, np.abs(pc[:, 2]) * 100])
    clear = margin > 1e-3
    assert (r["visible"][clear].astype(bool) == vis[clear]).all() and 0.02 * len(vis) < vis.sum() < 0.5 * len(vis)
    k = r["visible"].astype(bool)
    assert np.abs(r["uv"][k, 0] - u[k]).max() < 1e-3 and np.abs(r["distance"][k] - d[k]).max() < 1e-5
    assert np.abs(r["cos_theta"][k] - cos[k]).max() < 1e-6
    lvl = np.clip(np.rint(np.log(max_d[k].astype(np.float64) / d[k]) / np.log(1.2)), 0, 7)
    assert (r["level"][k] != lvl).mean() < 0.002 and set(np.unique(r["level"][k])) <= set(range(8))


@pytest.mark.gpu
def test_project_map_points_matches_restatement():
    from oracle import pyoracle
    from orb_slam2_ros2_amd._lib import Context
    orc = pyoracle.Oracle(pyoracle.build())
    ctx = Context(640, 480, 1000, 8, 1.2, 20, 7, max_images=1)
    for seed, n in ((5, 4000), (6, 1), (7, 257)):
        args = _vision_case(n, seed)
        got, want = ctx.project_map_points(*args), orc.project_map_points(*args)
        assert (got["visible"] == want["visible"]).all()
        k = want["visible"].astype(bool)
        for key in ("uv", "distance", "cos_theta", "level"):  # bit-exact float results where the reference defines them
            assert (got[key][k] == want[key][k]).all(), key
    empty = ctx.project_map_points(np.zeros((0, 3)), np.zeros((0, 3)), np.zeros(0), np.zeros(0), np.eye(3), np.zeros(3),
                                   (500, 500, 320, 240), (0, 640, 0, 480))
    assert len(empty["visible"]) == 0
    ctx.close()
