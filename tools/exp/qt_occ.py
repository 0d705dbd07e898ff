#!/usr/bin/env python3
"""Quadtree stage time of a 512-pair device batch at a given nFeatures (the LDS per tree follows the largest quota)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from orb_slam2_ros2_amd import synth
from orb_slam2_ros2_amd._lib import Context
nf = int(sys.argv[1]); n = 512
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1241
H = int(sys.argv[3]) if len(sys.argv) > 3 else 376
fr = [synth.stereo_pair(f, W, H) for f in range(16)]
dl = torch.from_numpy(np.stack([fr[i % 16][0] for i in range(n)])).cuda()
dr = torch.from_numpy(np.stack([fr[i % 16][1] for i in range(n)])).cuda()
ctx = Context(W, H, n_features=nf, max_images=2 * n)
for _ in range(3):
    ctx.stereo_batch_device(dl.data_ptr(), dr.data_ptr(), W, W * H, n, 718.856, 386.1)
ctx.sync()
ctx.profile_enable(1); ctx.profile_read()
for _ in range(10):
    ctx.stereo_batch_device(dl.data_ptr(), dr.data_ptr(), W, W * H, n, 718.856, 386.1)
st = ctx.profile_read()
print("size", W, H, "nFeatures", nf, {k: round(v[0] / 10, 4) for k, v in st.items() if v[1]})
ctx.close()
