#!/usr/bin/env python3
"""orbfe_ba_local_optimize over the window size: wall clock per call (host arrays in, results out) for a list of free-keyframe counts,
each checked against the CPU oracle (pose / point difference, iteration counts).  Usage: lba_sizes.py [nf ...] [--no-oracle]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_ros2_amd import ba_synth
from orb_slam2_ros2_amd._lib import Context
from oracle import pyoracle
args = [a for a in sys.argv[1:] if not a.startswith("--")]
sizes = [int(a) for a in args] or [30, 40, 43, 64, 100, 300]
check = "--no-oracle" not in sys.argv
ctx = Context(640, 480, n_features=500, max_images=1)
orc = pyoracle.Oracle(pyoracle.build(fast=True, out_dir="/tmp/orb_oracle_lba")) if check else None
for nf in sizes:
    n_fixed = 10
    n_kf = nf + n_fixed
    pr = ba_synth.make_problem(seed=100 + nf, n_kf=n_kf, n_pt=50 * n_kf, with_truth=True)
    fixed = np.zeros(n_kf, np.uint8); fixed[:n_fixed] = 1
    pr["poses"][:n_fixed] = pr["poses_true"][:n_fixed]
    g = ctx.ba_local_optimize(pr, fixed)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        g = ctx.ba_local_optimize(pr, fixed)
    ms = (time.perf_counter() - t0) / reps * 1e3
    line = f"free {nf:4d}  edges {len(pr['edge_pose']):6d}  device {ms:8.3f} ms  iters {tuple(g['iters'])}"
    if check:
        t2 = time.perf_counter(); o = orc.ba_local_optimize(pr, fixed); t3 = time.perf_counter()
        line += f"  cpu oracle {1e3 * (t3 - t2):8.1f} ms  iters {tuple(o['iters'])}  max pose diff {np.abs(g['poses'] - o['poses']).max():.2e}  point diff {np.abs(g['points'] - o['points']).max():.2e}"
    print(line, flush=True)
ctx.close()
