#!/usr/bin/env python3
"""Wall-clock per batched step (no profiling), for A/B experiments.  Usage: step_time.py [pairs] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orb_slam2_ros2_amd import synth
from orb_slam2_ros2_amd._lib import Context
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
prs = [synth.stereo_pair(i) for i in range(8)]
dl = torch.from_numpy(np.stack([prs[i % 8][0] for i in range(B)])).cuda()
dr = torch.from_numpy(np.stack([prs[i % 8][1] for i in range(B)])).cuda()
ctx = Context(1241, 376, max_images=2 * B)
for _ in range(3):
    ctx.stereo_batch_device(dl.data_ptr(), dr.data_ptr(), 1241, 1241 * 376, B, 718.856, 386.14)
ctx.sync()
t0 = time.perf_counter()
for _ in range(K):
    ctx.stereo_batch_device(dl.data_ptr(), dr.data_ptr(), 1241, 1241 * 376, B, 718.856, 386.14)
ctx.sync()
dt = time.perf_counter() - t0
nm = ctx.fetch_stereo(B - 1)[0]
print(f"pairs/step {B}  ms/step {dt / K * 1e3:.3f}  pairs/s {B * K / dt:.0f}  (matches of last pair {nm})")
