#!/usr/bin/env python3
"""Latency of the drop-in single-frame path (host images in, packed results out): extract_batch([L,R]) + stereo_match."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_ros2_amd import synth
from orb_slam2_ros2_amd._lib import Context
L, R = synth.stereo_pair(0)
ctx = Context(1241, 376, max_images=2)
for _ in range(20):
    ctx.extract_batch([L, R]); ctx.stereo_match(0, 1, 718.856, 386.14)
ts = []
for _ in range(200):
    t0 = time.perf_counter(); ctx.extract_batch([L, R]); t1 = time.perf_counter(); ctx.stereo_match(0, 1, 718.856, 386.14); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t1))
a = np.array(ts) * 1e3
print("extract_batch ms: median %.3f  p90 %.3f | stereo_match ms: median %.3f | pair total median %.3f ms" % (np.median(a[:, 0]), np.percentile(a[:, 0], 90), np.median(a[:, 1]), np.median(a.sum(1))))
ctx.profile_enable(True)
for _ in range(50):
    ctx.extract_batch([L, R]); ctx.stereo_match(0, 1, 718.856, 386.14)
p = ctx.profile_read()
print({k: round(ms / n, 4) for k, (ms, n) in p.items() if n}, "sum", round(sum(ms / n for ms, n in p.values() if n), 3))
# the reference's own call shape (Frame.cc:100-105): two extractor objects on two threads, each in its own slot (orbfe_extract_slot), then
# searchByStereo on the two slots
import threading
out = {}
def run(slot, img):
    out[slot] = ctx2.extract_slot(slot, img)
ctx2 = Context(1241, 376, max_images=4)
ts = []
for it in range(220):
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(s, im)) for s, im in ((0, L), (1, R))]
    [t.start() for t in th]; [t.join() for t in th]
    t1 = time.perf_counter()
    ctx2.stereo_match(0, 1, 718.856, 386.14)
    t2 = time.perf_counter()
    if it >= 20:
        ts.append((t1 - t0, t2 - t1))
a = np.array(ts) * 1e3
print("two threads x extract_slot ms: median %.3f  p90 %.3f | stereo_match ms: median %.3f | pair total median %.3f ms (python thread start/join included)" % (
    np.median(a[:, 0]), np.percentile(a[:, 0], 90), np.median(a[:, 1]), np.median(a.sum(1))))
