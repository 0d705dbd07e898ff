#!/usr/bin/env python3
"""One library variant on one content class: stage times with every kernel alone (HIP events) and the wall-clock step of the production
schedule, results digested so that two variants can be compared for equality.  Run on the GPU box, one process per variant:
    ORBFE_LIB=tools/exp/libs/liborbfe_x.so python tools/ab_content.py rect 512 60
(ORBFE_LIB: copied over orb_slam2_ros2_amd/liborbfe_hip.so by the caller -- see tools/exp/ab_content.sh)"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from orb_slam2_ros2_amd import synth
from orb_slam2_ros2_amd._lib import Context

cls = sys.argv[1] if len(sys.argv) > 1 else "rect"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
K = int(sys.argv[3]) if len(sys.argv) > 3 else 60
U = 8
prs = [synth.stereo_pair_content(i, cls) for i in range(U)]
dl = torch.from_numpy(np.stack([prs[i % U][0] for i in range(B)])).cuda()
dr = torch.from_numpy(np.stack([prs[i % U][1] for i in range(B)])).cuda()
ctx = Context(1241, 376, max_images=2 * B)


def step():
    ctx.stereo_batch_device(dl.data_ptr(), dr.data_ptr(), 1241, 1241 * 376, B, 718.856, 386.14)


t_end = time.perf_counter() + 1.0
while time.perf_counter() < t_end:
    step()
    ctx.sync()
ctx.profile_enable(1)
ctx.profile_read()
for _ in range(8):
    step()
ctx.sync()
alone = {k: ms / n for k, (ms, n) in ctx.profile_read().items() if n}
ctx.profile_enable(0)
for _ in range(3):
    step()
ctx.sync()
t0 = time.perf_counter()
for _ in range(K):
    step()
ctx.sync()
ms = (time.perf_counter() - t0) / K * 1e3
kps, desc, cnt = ctx.fetch_batch(0, 2 * min(B, 16))
ru, dp, nm = ctx.fetch_stereo_batch(0, min(B, 16))
h = hashlib.sha256()
for a in (kps, desc, cnt, ru, dp, nm):
    h.update(np.ascontiguousarray(a).tobytes())
print(f"{cls:10s} B {B}  step {ms:.3f} ms  " + "  ".join(f"{k} {v:.3f}" for k, v in alone.items()) + f"  digest {h.hexdigest()[:12]}")
