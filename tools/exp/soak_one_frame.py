#!/usr/bin/env python3
"""Soak of the one-frame paths (r6: cooperative quadtree steps, sharded lists, eight-part row table with flag waits): N frame_stereo calls on
changing frames, every result hashed against the first call on the same frame; then extract_slot on two threads.  tools/exp/soak_one_frame.py [N]"""
import hashlib, sys, threading, time
sys.path.insert(0, ".")
import numpy as np
from orb_slam2_ros2_amd import synth
from orb_slam2_ros2_amd._lib import Context
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
FX, BF = 718.856, 718.856 * 0.537166
frames = [synth.stereo_pair(f) for f in range(6)]
ctx = Context(1241, 376, max_images=2)
def dig(res):
    (lk, ld), (rk, rd), nm, ru, dp = res
    h = hashlib.sha256()
    for a in (lk, ld, rk, rd, ru, dp): h.update(np.ascontiguousarray(a).tobytes())
    h.update(str(nm).encode())
    return h.hexdigest()
ref = {}
t0 = time.time()
for i in range(N):
    f = (i * 5 + i // 7) % len(frames)
    d = dig(ctx.frame_stereo(frames[f][0], frames[f][1], FX, BF))
    if f not in ref: ref[f] = d
    assert ref[f] == d, f"call {i} frame {f}: digest changed"
print(f"frame_stereo: {N} calls on {len(frames)} frames, all digests stable, {time.time() - t0:.1f} s")
ctx.close()
ctx = Context(1241, 376, max_images=2)
ref2 = {}
out = [None, None]
def work(slot, img):
    out[slot] = ctx.extract_slot(slot, img)
t0 = time.time()
for i in range(N // 5):
    f = i % len(frames)
    th = [threading.Thread(target=work, args=(s, frames[f][s])) for s in (0, 1)]
    for t in th: t.start()
    for t in th: t.join()
    nm, ru, dp, _, _ = ctx.stereo_match(0, 1, FX, BF)
    d = dig((out[0], out[1], nm, ru, dp))
    if f not in ref2: ref2[f] = d
    assert ref2[f] == d, f"threads round {i} frame {f}: digest changed"
    assert ref2[f] == ref[f], "two threads differ from the one call"
print(f"two threads + match: {N // 5} rounds, all digests stable and equal to the one call, {time.time() - t0:.1f} s")
ctx.close()
