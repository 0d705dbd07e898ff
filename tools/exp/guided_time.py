#!/usr/bin/env python3
"""Host-call latency of the guided-matcher entry points at a frame's size (2000 target features, 1000 queries / map points)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from orb_slam2_ros2_amd import synth
from orb_slam2_ros2_amd._lib import Context
L, R = synth.stereo_pair(3)
ctx = Context(1241, 376, max_images=4)
(lk, ld), (rk, rd) = ctx.extract_batch([L, R])
r = np.random.default_rng(0)
nq = 1000
q = r.integers(0, len(rk), nq)
qxy = np.stack([rk["x"][q], rk["y"][q]], 1).astype(np.float32) + r.normal(0, 3, (nq, 2)).astype(np.float32)
rad = r.uniform(5, 40, nq).astype(np.float32)
lo, hi = np.zeros(nq, np.int8), np.full(nq, 7, np.int8)
def med(f, n=200):
    for _ in range(10): f()
    ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts))
print("search_in_area (slot-resident target)   %.3f ms" % med(lambda: ctx.search_in_area(0, qxy, rad, lo, hi, rd[q])))
print("search_in_area_features (uploaded)      %.3f ms" % med(lambda: ctx.search_in_area_features(lk, ld, qxy, rad, lo, hi, rd[q])))
pos = r.uniform(-5, 5, (2000, 3)).astype(np.float32); pos[:, 2] = r.uniform(3, 30, 2000)
vd = np.tile(np.array([0, 0, 1], np.float32), (2000, 1))
mx, mn = np.full(2000, 100, np.float32), np.full(2000, 0.1, np.float32)
print("project_map_points (2000)               %.3f ms" % med(lambda: ctx.project_map_points(pos, vd, mx, mn, np.eye(3), np.zeros(3), (718.856, 718.856, 607.19, 185.2), (0, 1241, 0, 376))))
print("match_bruteforce 1000 x 2000            %.3f ms" % med(lambda: ctx.match_bruteforce(rd[q], ld)))
ctx.close()
