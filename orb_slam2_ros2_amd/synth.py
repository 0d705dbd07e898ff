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


def _draw_scene(seed, w, h, n_rect, right, sparse):
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
