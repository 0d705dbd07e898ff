"""Known-answer tests for the oracle's cv::FAST restatement and the Quadtree selection
(ORBExtractor.cc:19-192): hand-made patches, an independent threshold-free formulation (the one the HIP
kernel uses, proved equivalent in DESIGN.md), and an independent heap-based quadtree."""
import heapq

import numpy as np
import pytest

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0),
        (-3, 1), (-2, 2), (-1, 3)]


def v_map(patch):
    """Threshold-free V = max over the 16 arcs of 9 ring pixels of min(+-d); cornerScore = V - 1, corner at t iff V > t."""
    h, w = patch.shape
    V = np.full((h, w), -1000, np.int64)
    p = patch.astype(np.int64)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            d = np.array([p[y, x] - p[y + dy, x + dx] for dx, dy in RING])
            best = -1000
            for s in range(16):
                idx = [(s + k) % 16 for k in range(9)]
                best = max(best, d[idx].min(), (-d[idx]).min())
            V[y, x] = best
    return V


def fast_by_definition(patch, t):
    V = v_map(patch)
    h, w = patch.shape
    S = np.where(V > t, V, 0)  # non-corners score 0, outside the interior 0
    S[:3] = S[-3:] = 0
    S[:, :3] = S[:, -3:] = 0
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            v = S[y, x]
            if v <= 0:
                continue
            nb = [S[y + j, x + i] for j in (-1, 0, 1) for i in (-1, 0, 1) if (i, j) != (0, 0)]
            if all(v > q for q in nb):
                out.append((x, y, v - 1))
    return np.asarray(out, np.int32).reshape(-1, 3)


def ring_patch(center, ring_vals, size=7):
    p = np.full((size, size), center, np.uint8)
    c = size // 2
    for (dx, dy), v in zip(RING, ring_vals):
        p[c + dy, c + dx] = v
    return p


def test_fast_bright_arc_of_9_is_a_corner_arc_of_8_is_not(orc):
    nine = ring_patch(100, [150] * 9 + [100] * 7)
    assert orc.fast(nine, 20).tolist() == [[3, 3, 49]]      # score = largest t still a corner = 50-1
    assert orc.fast(nine, 49).tolist() == [[3, 3, 49]]
    assert len(orc.fast(nine, 50)) == 0
    eight = ring_patch(100, [150] * 8 + [100] * 8)
    assert len(orc.fast(eight, 20)) == 0
    wrap = ring_patch(100, [60] * 4 + [100] * 7 + [60] * 5)     # dark arc wrapping around index 0
    assert orc.fast(wrap, 20).tolist() == [[3, 3, 39]]


def test_fast_score_is_min_over_the_best_arc(orc):
    vals = [150, 160, 170, 125, 180, 190, 150, 150, 150] + [100] * 7
    assert orc.fast(ring_patch(100, vals), 20).tolist() == [[3, 3, 24]]


def test_fast_threshold_is_clamped_and_strict(orc):
    p = ring_patch(100, [121] * 9 + [100] * 7)
    assert len(orc.fast(p, 20)) == 1 and len(orc.fast(p, 21)) == 0  # needs ring > v + t strictly
    assert len(orc.fast(p, -5)) == 1                                  # clamped to 0


def test_fast_nms_ties_suppress_both_and_patch_edges_do_not_see_outside(orc, rng):
    # two identical corners side by side: equal scores, strict '>' => neither survives
    p = np.full((9, 12), 100, np.uint8)
    for cx in (4, 5):
        for (dx, dy) in RING[:9]:
            p[4 + dy, cx + dx] = 200
    by_def = fast_by_definition(p, 20)
    got = orc.fast(p, 20)
    assert np.array_equal(got, by_def)
    # the same image cut into a narrower patch changes what NMS can see (per-patch semantics)
    sub = p[:, :10]
    assert np.array_equal(orc.fast(sub, 20), fast_by_definition(sub, 20))


@pytest.mark.parametrize("seed", range(6))
def test_fast_matches_threshold_free_definition_on_random_patches(orc, seed):
    r = np.random.default_rng(seed)
    base = r.integers(0, 256, (6, 7)).astype(np.uint8)
    patch = np.kron(base, np.ones((6, 6), np.uint8))[:33, :38]  # blocky image: many real corners
    patch = np.clip(patch.astype(int) + r.integers(-4, 5, patch.shape), 0, 255).astype(np.uint8)
    for t in (7, 20, 40):
        assert np.array_equal(orc.fast(patch, t), fast_by_definition(patch, t)), (seed, t)


# ---------------------------------------------------------------------------------------------------
def py_quadtree(w, h, pts, need):
    """Independent restatement: heap ordered by (-count, insertion seq) == multimap<count, greater> order."""
    pts = [(float(x), float(y), float(r)) for x, y, r in pts]

    def inside(b, idxs):
        rb, re, cb, ce = b
        return [i for i in idxs if cb < pts[i][0] < ce and rb < pts[i][1] < re]

    seq = 0
    heap = []
    root_idx = list(range(len(pts)))
    n_ini = int(np.floor(w / h + 0.5)) if w / h >= 0 else 0
    hx = np.float32(np.float64(w) / n_ini) if n_ini else np.float32(0)
    cols = [0.0] + [float(np.float32(i) * hx) for i in range(1, n_ini)] + [float(w)]
    n_nodes = 1
    live = [(-len(root_idx), -1, "root")]
    if not (n_nodes < need):
        if need >= 1 and pts:
            best = max(range(len(pts)), key=lambda i: (pts[i][2], -i))
            return [best]
        return []
    # pop the root
    n_nodes -= 1
    for i in range(n_ini):
        b = (0.0, float(h), cols[i], cols[i + 1])
        idxs = inside(b, root_idx)
        if idxs:
            heapq.heappush(heap, (-len(idxs), seq, b, idxs))
            seq += 1
            n_nodes += 1
    while n_nodes < need and heap:
        _, _, (rb, re, cb, ce), idxs = heapq.heappop(heap)
        n_nodes -= 1
        mr, mc = (rb + re) / 2, (cb + ce) / 2
        for b in ((rb, mr, cb, mc), (rb, mr, mc, ce), (mr, re, cb, mc), (mr, re, mc, ce)):
            sub = inside(b, idxs)
            if sub:
                heapq.heappush(heap, (-len(sub), seq, b, sub))
                seq += 1
                n_nodes += 1
    ordered = sorted(heap)[:need]
    out = set()
    for _, _, _, idxs in ordered:
        best, br = idxs[0], 0.0
        first = True
        for i in idxs:
            if pts[i][2] > br:
                best, br = i, pts[i][2]
        out.add(best)
    return sorted(out)


def test_quadtree_four_quadrants(orc):
    pts = [(10, 10, 5), (90, 10, 6), (10, 90, 7), (90, 90, 8)]
    sel, splits = orc.quadtree(100, 100, np.asarray(pts, np.float32), 4)
    assert sel.tolist() == [0, 1, 2, 3] and splits == 2  # root pop + one split


def test_quadtree_points_on_split_lines_are_dropped(orc):
    pts = [(50, 10, 9), (10, 50, 9), (10, 10, 5), (90, 90, 6), (90, 10, 7), (10, 90, 8)]
    sel, _ = orc.quadtree(100, 100, np.asarray(pts, np.float32), 4)
    assert sel.tolist() == [2, 3, 4, 5]  # x==50 / y==50 lie on the first split and vanish (ORBExtractor.h:55-62)


def test_quadtree_picks_first_maximum_response_per_node(orc):
    pts = [(10, 10, 5), (12, 12, 9), (14, 14, 9), (90, 90, 1)]
    sel, _ = orc.quadtree(100, 100, np.asarray(pts, np.float32), 2)
    assert sel.tolist() == [1, 3]  # node {0,1,2}: strict '>' keeps the first 9


def test_quadtree_fewer_candidates_than_quota_returns_nothing(orc):
    r = np.random.default_rng(3)
    pts = np.stack([r.integers(3, 1200, 40), r.integers(3, 340, 40), r.integers(7, 200, 40)], 1).astype(np.float32)
    sel, splits = orc.quadtree(1209, 344, pts, 434)
    assert len(sel) == 0 and splits > 40 * 30  # quirk Q3: every point is eventually hit by a midpoint


def test_quadtree_truncates_to_quota_dropping_smallest_nodes(orc):
    # one strip, first split gives counts (3,1,1,1) -> 4 nodes >= need 3 -> keep the 3 first in (count desc, insertion) order
    pts = [(10, 10, 5), (20, 20, 6), (30, 30, 7), (90, 10, 8), (10, 90, 9), (90, 90, 10)]
    sel, _ = orc.quadtree(100, 100, np.asarray(pts, np.float32), 3)
    assert sel.tolist() == [2, 3, 4]  # TL node -> idx 2 (max response), then TR, BL; BR is truncated


def test_quadtree_quota_zero_and_one(orc):
    pts = np.asarray([(10, 10, 5), (20, 20, 9), (30, 30, 9)], np.float32)
    assert orc.quadtree(100, 100, pts, 0)[0].tolist() == []
    assert orc.quadtree(100, 100, pts, 1)[0].tolist() == [1]  # root->getFeature(): first max


@pytest.mark.parametrize("seed,n,need,w,h", [(0, 3000, 434, 1209, 344), (1, 500, 434, 1209, 344), (2, 800, 119, 314, 73),
                                             (3, 1500, 217, 608, 448), (4, 60, 30, 100, 300), (5, 700, 50, 1000, 100)])
def test_quadtree_matches_independent_heap_implementation(orc, seed, n, need, w, h):
    r = np.random.default_rng(seed)
    pts = np.stack([r.integers(3, w - 3, n), r.integers(3, h - 3, n), r.integers(7, 120, n)], 1).astype(np.float32)
    sel, _ = orc.quadtree(w, h, pts, need)
    assert sel.tolist() == py_quadtree(w, h, pts.tolist(), need)
