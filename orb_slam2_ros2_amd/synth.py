"""Synthetic KITTI-/TUM-shaped inputs (there is no dataset on either box).

Integer-only, counter-based (splitmix64 of (seed, stream, index)), so every value is a pure function
of its coordinates: numpy can vectorise it and any other language can reproduce it bit-for-bit.

Scene model (per frame index f): a slowly varying background with 16x16 blocks of random gray,
then `n_rect` axis-aligned rectangles drawn far-to-near; a rectangle is either flat or carries a
checkerboard texture; the right image draws the same scene with each layer shifted left by its
integer disparity; finally independent per-pixel noise in [-3, 3] on each image.  This gives several
times more FAST corners than the per-level quota on every pyramid level (so quirk Q3 -- a level with
fewer candidates than its quota returns nothing -- is not triggered by accident; tests trigger it on
purpose with `sparse=True`).
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + _GOLD
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return x


def hash_u64(seed: int, stream: int, idx):
    """h(seed, stream, idx) -> uint64; idx may be an array."""
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = splitmix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) ^ (np.uint64(stream) * np.uint64(0xD1B54A32D192ED03)))
        return splitmix64(base ^ (idx * _GOLD))


def _rand_int(seed, stream, idx, lo, hi):
    """uniform-ish integers in [lo, hi] (inclusive)."""
    span = np.uint64(hi - lo + 1)
    return (hash_u64(seed, stream, idx) % span).astype(np.int64) + lo


def frame_seed(f: int) -> int:
    return (0x00C0FFEE ^ (f * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF


def _background(seed, w, h, shift):
    ys, xs = np.mgrid[0:h, 0:w]
    xs = xs + shift
    bx = (xs // 16).astype(np.uint64)
    by = (ys // 16).astype(np.uint64)
    blk = (hash_u64(seed, 1, by * np.uint64(4096) + bx) % np.uint64(96)).astype(np.int64)
    grad = (xs * 40) // (w + 160) + (ys * 30) // h
    return 64 + blk + grad


def _draw_scene(seed, w, h, n_rect, right, sparse, big=False):
    img = _background(seed, w, h, 5 if right else 0)
    if sparse:
        img[:] = 96
    idx = np.arange(n_rect)
    rw = _rand_int(seed, 10, idx, 6, 120)
    rh = _rand_int(seed, 11, idx, 6, 90)
    rx = _rand_int(seed, 12, idx, -20, w)
    ry = _rand_int(seed, 13, idx, -20, h)
    g0 = _rand_int(seed, 14, idx, 16, 240)
    g1 = _rand_int(seed, 15, idx, 16, 240)
    tex = _rand_int(seed, 16, idx, 0, 2)       # 0: flat, 1: two-tone checker, 2: random mosaic
    cell = _rand_int(seed, 17, idx, 4, 18)
    if big:                                    # content class "sparse": large flat rectangles / coarse checkers only
        rw, rh = rw + 30, rh + 24
        tex = np.minimum(tex, 1)
        cell = cell * 3 + 24
    disp = _rand_int(seed, 18, idx, 6, 90)
    order = np.argsort(disp, kind="stable")    # far (small disparity) first, near last
    for i in order:
        x0 = int(rx[i]) - (int(disp[i]) if right else 0)
        y0 = int(ry[i])
        x1, y1 = x0 + int(rw[i]), y0 + int(rh[i])
        cx0, cy0, cx1, cy1 = max(x0, 0), max(y0, 0), min(x1, w), min(y1, h)
        if cx0 >= cx1 or cy0 >= cy1:
            continue
        if tex[i] == 0:
            img[cy0:cy1, cx0:cx1] = g0[i]
        else:
            yy, xx = np.mgrid[cy0:cy1, cx0:cx1]
            cxi = (xx - x0) // int(cell[i])
            cyi = (yy - y0) // int(cell[i])
            if tex[i] == 1:
                img[cy0:cy1, cx0:cx1] = np.where(((cxi + cyi) & 1) == 0, g0[i], g1[i])
            else:
                key = (cyi * 64 + cxi).astype(np.uint64) + np.uint64(int(i) << 16)
                img[cy0:cy1, cx0:cx1] = 16 + (hash_u64(seed, 19, key) % np.uint64(225)).astype(np.int64)
    return img


def stereo_pair(f: int, w: int = 1241, h: int = 376, n_rect: int = 260, sparse: bool = False):
    """Return (left, right) uint8 images of frame index f."""
    seed = frame_seed(f)
    out = []
    for right in (False, True):
        img = _draw_scene(seed, w, h, n_rect if not sparse else 6, right, sparse)
        pix = np.arange(w * h, dtype=np.uint64).reshape(h, w)
        noise = (hash_u64(seed, 30 + int(right), pix) % np.uint64(7)).astype(np.int64) - 3
        out.append(np.clip(img + noise, 0, 255).astype(np.uint8))
    return out[0], out[1]


def mono_image(f: int, w: int = 640, h: int = 480, n_rect: int = 300):
    """TUM-shaped gray image (config 5)."""
    return stereo_pair(f, w, h, n_rect)[0]


def descriptors_cfg3(n: int = 2000, seed_train: int = 1234, seed_query: int = 5678):
    """BASELINE config 3: n random 256-bit train descriptors; query i = train[perm(i)] with 0..40 bit
    flips (90 %) or a fresh random descriptor (10 %)."""
    idx = np.arange(n * 4, dtype=np.uint64)
    train = hash_u64(seed_train, 1, idx).view(np.uint8).reshape(n, 32).copy()
    perm = np.argsort(hash_u64(seed_query, 2, np.arange(n)), kind="stable")
    query = train[perm].copy()
    fresh = hash_u64(seed_query, 3, idx).view(np.uint8).reshape(n, 32)
    kind = hash_u64(seed_query, 4, np.arange(n)) % np.uint64(10)
    nflip = (hash_u64(seed_query, 5, np.arange(n)) % np.uint64(41)).astype(np.int64)
    for i in range(n):
        if kind[i] == 0:
            query[i] = fresh[i]
        else:
            bits = (hash_u64(seed_query, 6, np.arange(int(nflip[i])) + i * 64) % np.uint64(256)).astype(np.int64)
            for b in bits:
                query[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return query, train


# ---- content classes -----------------------------------------------------------------------------------------
# The throughput of the path depends on the image content: how many cells come out empty at the high FAST threshold and repeat at the
# low one (ORBExtractor.cc:365-367), how many candidates the quadtree has to spread, how many right keypoints share a row band.  The
# reference's input contract is a camera image (example/Stereo/KittiStereo.cc:28-33, RGB-D/TUMRGBD.cc:30); there is no dataset here,
# so four integer-only content classes stand in for the range:
#   "rect"      the rectangles + checkers + noise of stereo_pair() above (the class every earlier number was quoted on)
#   "camera"    1/f^2-like value noise with a smooth sky band and a smooth road, a few textured "facades", sensor noise sigma ~ 2.8:
#               a large share of the cells has no corner at 20 and takes the pass at 7 (many of them come out empty there too)
#   "saturated" hierarchical random mosaics: corners in every cell on every level, several times the quota
#   "sparse"    few large flat / coarse-checker rectangles on a flat background: roughly the same number of corners on every level, so
#               the fine levels stay below their quota and return nothing (quirk Q3), the coarse ones select
CONTENT_CLASSES = ("rect", "camera", "saturated", "sparse")


def _value_noise(seed, stream, xs, ys, spacing):
    """integer bilinear interpolation of hashed lattice values in [-128, 127] (lattice pitch `spacing` px); xs / ys are int64 grids >= 0"""
    gx, gy = xs // spacing, ys // spacing
    fx, fy = xs - gx * spacing, ys - gy * spacing

    def node(ix, iy):
        return (hash_u64(seed, stream, (iy.astype(np.uint64) << np.uint64(20)) + ix.astype(np.uint64)) % np.uint64(256)).astype(np.int64) - 128
    v00, v10, v01, v11 = node(gx, gy), node(gx + 1, gy), node(gx, gy + 1), node(gx + 1, gy + 1)
    top = v00 * (spacing - fx) + v10 * fx
    bot = v01 * (spacing - fx) + v11 * fx
    return (top * (spacing - fy) + bot * fy) // (spacing * spacing)


def _camera_scene(seed, w, h, right):
    ys, xs = np.mgrid[0:h, 0:w].astype(np.int64)
    # ground-plane-like disparity: 2 px above the horizon, growing linearly below it; the right image samples the scene at x + d(y)
    hor = (h * 2) // 5
    disp_row = np.where(np.arange(h) < hor, 2, 2 + ((np.arange(h) - hor) * 70) // max(1, h - hor)).astype(np.int64)
    xs = xs + 256 + (disp_row[:, None] if right else 0)
    # 1/f^2-like texture: octaves of value noise, amplitude proportional to the wavelength
    tex = np.zeros((h, w), np.int64)
    for k, (sp, amp) in enumerate(((128, 40), (64, 28), (32, 20), (16, 14), (8, 10), (4, 7), (2, 5))):
        tex += _value_noise(seed, 40 + k, xs, ys, sp) * amp
    tex = tex // 64                                            # about +-120 at full contrast
    # contrast mask: sky (above an undulating horizon) and road (a trapezoid at the bottom) are smooth, the middle band is textured
    hx = hor + _value_noise(seed, 50, xs, np.zeros_like(xs), 96) // 6        # horizon height per column, +-21 px
    sky = ys < hx
    cx_road = w // 2 + 256
    half = 40 + ((ys - hor) * (w // 2)) // max(1, h - hor)                   # road half-width grows towards the bottom
    road = (~sky) & (np.abs(xs - cx_road) < half) & (ys > hor + 12)
    img = np.where(sky, 205 - (ys * 50) // max(1, hor) + tex // 24,
                   np.where(road, 92 + tex // 12, 110 + tex))
    # lane dashes on the road (a few strong corners in an otherwise smooth region)
    dash = road & (np.abs(xs - cx_road) < 3 + (ys - hor) // 40) & ((((ys - hor) * (ys - hor)) // 64) % 2 == 0)
    img = np.where(dash, 225, img)
    # facades: textured rectangles with window grids in the middle band, each at its own disparity (drawn far to near)
    idx = np.arange(14)
    rw = _rand_int(seed, 60, idx, 60, 220)
    rh = _rand_int(seed, 61, idx, 40, 120)
    rx = _rand_int(seed, 62, idx, 0, w)
    ry = _rand_int(seed, 63, idx, hor - 90, hor + 10)
    g0 = _rand_int(seed, 64, idx, 60, 190)
    cell = _rand_int(seed, 65, idx, 9, 22)
    disp = _rand_int(seed, 66, idx, 4, 40)
    for i in np.argsort(disp, kind="stable"):
        x0 = int(rx[i]) - (int(disp[i]) if right else 0)
        y0 = int(ry[i])
        cx0, cy0, cx1, cy1 = max(x0, 0), max(y0, 0), min(x0 + int(rw[i]), w), min(y0 + int(rh[i]), h)
        if cx0 >= cx1 or cy0 >= cy1:
            continue
        yy, xx = np.mgrid[cy0:cy1, cx0:cx1]
        u, v = (xx - x0) % int(cell[i]), (yy - y0) % int(cell[i])
        win = (u >= 2) & (u < int(cell[i]) - 3) & (v >= 2) & (v < int(cell[i]) - 2)
        img[cy0:cy1, cx0:cx1] = np.where(win, int(g0[i]) - 45, int(g0[i])) + tex[cy0:cy1, cx0:cx1] // 16
    return img


def _saturated_scene(seed, w, h, right):
    ys, xs = np.mgrid[0:h, 0:w].astype(np.int64)
    band = ys // 47
    d = 8 + 9 * (band % 5)                                    # depth layers in horizontal bands
    xs = xs + 128 + (d if right else 0)

    def mosaic(stream, pitch, span):
        key = ((ys // pitch).astype(np.uint64) << np.uint64(20)) + (xs // pitch).astype(np.uint64)
        return (hash_u64(seed, stream, key) % np.uint64(2 * span + 1)).astype(np.int64) - span
    return 128 + mosaic(70, 48, 40) + mosaic(71, 21, 36) + mosaic(72, 9, 30) + mosaic(73, 4, 18)


def stereo_pair_content(f: int, content: str = "rect", w: int = 1241, h: int = 376):
    """(left, right) uint8 images of frame index f of one of CONTENT_CLASSES; "rect" is stereo_pair(f, w, h)."""
    if content == "rect":
        return stereo_pair(f, w, h)
    if content not in CONTENT_CLASSES:
        raise ValueError(f"content class {content!r}: one of {CONTENT_CLASSES}")
    seed = frame_seed(f) ^ {"camera": 0xCA3E4A, "saturated": 0x5A7024, "sparse": 0x59A45E}[content]
    out = []
    for right in (False, True):
        if content == "camera":
            img = _camera_scene(seed, w, h, right)
        elif content == "saturated":
            img = _saturated_scene(seed, w, h, right)
        else:
            img = _draw_scene(seed, w, h, 70, right, True, big=True)
        pix = np.arange(w * h, dtype=np.uint64).reshape(h, w)
        n1 = (hash_u64(seed, 30 + int(right), pix) % np.uint64(7)).astype(np.int64) - 3
        if content == "camera":   # sensor noise: two uniform draws, sigma ~ 2.8
            n1 = n1 + (hash_u64(seed, 32 + int(right), pix) % np.uint64(7)).astype(np.int64) - 3
        elif content == "sparse":  # +-1: a flat region stays below the low threshold
            n1 = (hash_u64(seed, 30 + int(right), pix) % np.uint64(3)).astype(np.int64) - 1
        out.append(np.clip(img + n1, 0, 255).astype(np.uint8))
    return out[0], out[1]


def lo_pass_cells(cands, w, h, th_hi=20):
    """(cells that took the low-threshold pass, cells) of one pyramid level, from its FAST candidate records (x, y in the level's
    bordered frame of w x h = level size - 32, response = cornerScore): the grid of ORBExtractor.cc:346-363 (30-px cells, integer
    division first); a cell repeats cv::FAST at the low threshold iff it holds no corner at th_hi (:365-367), i.e. no candidate whose
    score reaches th_hi.  Host-side analysis of a result, used by the content statistics and bench.py's content sweep."""
    n_cols, n_rows = w // 30, h // 30
    if n_cols <= 0 or n_rows <= 0:
        return 0, 0
    wc, hc = w // n_cols, h // n_rows
    live = sum(1 for j in range(n_cols) if j * wc < w - 6) * sum(1 for i in range(n_rows) if i * hc < h - 6)
    c = np.asarray(cands)
    if len(c) == 0:
        return live, live
    hi = c[c[:, 2] >= th_hi]
    cy = np.minimum((hi[:, 1].astype(np.int64) - 3) // hc, n_rows - 1)
    cx = np.minimum((hi[:, 0].astype(np.int64) - 3) // wc, n_cols - 1)
    return live - len(np.unique(cy * n_cols + cx)), live
