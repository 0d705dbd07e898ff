#!/usr/bin/env python3
"""Wall-clock of orbfe_ba_local_optimize on BASELINE config 5's synthetic local map against the CPU oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_ros2_amd import ba_synth
from orb_slam2_ros2_amd._lib import Context
from oracle import pyoracle
pr = ba_synth.make_problem(seed=42, n_kf=60, n_pt=3000, with_truth=True)
fixed = np.zeros(60, np.uint8); fixed[:20] = 1
pr["poses"][:20] = pr["poses_true"][:20]
ctx = Context(640, 480, n_features=500, max_images=1)
ctx.ba_local_optimize(pr, fixed)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
t0 = time.perf_counter()
for _ in range(reps):
    g = ctx.ba_local_optimize(pr, fixed)
t1 = t0 + (time.perf_counter() - t0) / reps
orc = pyoracle.Oracle(pyoracle.build(fast=True, out_dir="/tmp/orb_oracle_lba"))
t2 = time.perf_counter(); o = orc.ba_local_optimize(pr, fixed); t3 = time.perf_counter()
print(f"edges {len(pr['edge_pose'])} free poses 40: device {1e3 * (t1 - t0):.3f} ms, cpu oracle {1e3 * (t3 - t2):.1f} ms, iters {g['iters']}, "
      f"max pose diff {np.abs(g['poses'] - o['poses']).max():.2e}")
