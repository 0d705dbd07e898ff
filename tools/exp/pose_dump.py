#!/usr/bin/env python3
"""Writes the pose-only problem of the bench's `ba` leg (ba_synth.make_pose_problem) to a binary file for tools/exp/pose_bench.hip."""
import os, struct, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from orb_slam2_ros2_amd import ba_synth
pp = ba_synth.make_pose_problem()
n = len(pp["info"])
with open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/pose.bin", "wb") as f:
    f.write(struct.pack("i", n))
    f.write(np.array([pp["fx"], pp["fy"], pp["cx"], pp["cy"], pp["bf"]], np.float64).tobytes())
    f.write(np.asarray(pp["pose"], np.float64).tobytes())
    for k, t in (("Xw", np.float64), ("meas", np.float64), ("info", np.float64), ("sigma2", np.float32)):
        f.write(np.ascontiguousarray(pp[k], t).tobytes())
print(n)
