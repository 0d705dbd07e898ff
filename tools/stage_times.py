#!/usr/bin/env python3
"""Print per-stage device time (HIP events) for one batched step; no result checks (used for phase experiments)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orb_slam2_ros2_amd import synth
from orb_slam2_ros2_amd._lib import Context
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prs = [synth.stereo_pair(i) for i in range(8)]
dl = torch.from_numpy(np.stack([prs[i % 8][0] for i in range(B)])).cuda()
dr = torch.from_numpy(np.stack([prs[i % 8][1] for i in range(B)])).cuda()
ctx = Context(1241, 376, max_images=2 * B)
for _ in range(3):
    ctx.stereo_batch_device(dl.data_ptr(), dr.data_ptr(), 1241, 1241 * 376, B, 718.856, 386.14)
ctx.sync(); ctx.profile_enable(True)
for _ in range(5):
    ctx.stereo_batch_device(dl.data_ptr(), dr.data_ptr(), 1241, 1241 * 376, B, 718.856, 386.14)
ctx.sync()
p = ctx.profile_read()
print({k: round(ms / n, 4) for k, (ms, n) in p.items() if n}, "total", round(sum(ms / n for ms, n in p.values() if n), 3))
