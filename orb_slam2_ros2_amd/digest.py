"""Digest of one stereo pair's results (what Frame::createStereo leaves in a Frame): used by the golden fixtures, the GPU tests and
bench.py's after-the-clock verification.  Plain hashlib over the exact bytes."""
from __future__ import annotations

import hashlib

import numpy as np


def pair_digest(lk, ld, rk, rd, right_u, depth, n_matches) -> str:
    """sha256 over keypoints(L) | descriptors(L) | keypoints(R) | descriptors(R) | right_u[:nL] | depth[:nL] | n_matches"""
    n = len(lk)
    h = hashlib.sha256()
    for a in (lk, ld, rk, rd, np.asarray(right_u)[:n], np.asarray(depth)[:n]):
        h.update(np.ascontiguousarray(a).tobytes())
    h.update(int(n_matches).to_bytes(4, "little", signed=True))
    return h.hexdigest()


def batch_digests(kps, desc, counts, right_u, depth, n_matches):
    """digests of every pair of a packed batch: kps [2P][NF], desc [2P][NF][32], counts [2P], right_u / depth [P][NF], n_matches [P]"""
    out = []
    for p in range(len(n_matches)):
        nl, nr = int(counts[2 * p]), int(counts[2 * p + 1])
        out.append(pair_digest(kps[2 * p, :nl], desc[2 * p, :nl], kps[2 * p + 1, :nr], desc[2 * p + 1, :nr], right_u[p], depth[p],
                               n_matches[p]))
    return out
