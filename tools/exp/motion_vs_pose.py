#!/usr/bin/env python3
"""Diagnostic: the pose-only optimisation inside orbfe_track_motion_model against orbfe_pose_only_optimize on the edge list it built."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from orb_slam2_ros2_amd import synth
from orb_slam2_ros2_amd._lib import Context
W, H, NF = 1241, 376, 2000
FX, BF, CX, CY = 718.856, 718.856 * 0.537166, 607.1928, 185.2157
L, R = synth.stereo_pair(0)
ctx = Context(W, H, max_images=2)
(lk, ld), _ = ctx.extract_batch([L, R])
nm, ru, dp, _, _ = ctx.stereo_match(0, 1, FX, BF)
r = np.random.default_rng(0)
n = len(lk)
ru_full = np.full(NF, -1.0); ru_full[:n] = ru[:n]
depth = np.where(dp[:n] > 0, dp[:n], r.uniform(4, 30, n))
X = np.stack([(lk["x"] - CX) / FX * depth, (lk["y"] - CY) / FX * depth, depth], 1).astype(np.float32)
qi = np.sort(r.permutation(n)[:1600])
qxy = np.stack([lk["x"][qi], lk["y"][qi]], 1).astype(np.float32) + r.normal(0, 3, (len(qi), 2)).astype(np.float32)
octv = lk["octave"][qi].astype(np.int8)
lo, hi = np.maximum(0, octv - 1).astype(np.int8), np.minimum(7, octv + 1).astype(np.int8)
sf = np.array([np.float32(1.2) ** l for l in range(8)], np.float32); sig2 = sf * sf; isig2 = (np.float32(1) / sig2).astype(np.float32)
p0 = np.array([0, 0, 0, 1, 0.03, -0.02, 0.04], np.float64)
g = ctx.track_motion_model(0, qxy, octv, lo, hi, ld[qi], X[qi], (FX, FX, CX, CY, BF), (0.0, float(W), 0.0, float(H)), p0, sig2, isig2, right_u=ru_full)
ef = np.flatnonzero(g["assigned"] >= 0)
meas = np.stack([lk["x"][ef].astype(np.float64), lk["y"][ef].astype(np.float64), ru_full[ef]], 1)
oc = lk["octave"][ef]
ng, pose, inl = ctx.pose_only_optimize(X[qi][g["assigned"][ef]].astype(np.float64), meas, isig2[oc].astype(np.float64), sig2[oc], p0, FX, FX, CX, CY, BF)
print("edges", len(ef), "n_good", g["n_good"], ng, "pose bitwise equal", np.array_equal(pose, g["pose"]), np.abs(pose - g["pose"]).max())
ctx.profile_enable(True)
for _ in range(20):
    ctx.track_motion_model(0, qxy, octv, lo, hi, ld[qi], X[qi], (FX, FX, CX, CY, BF), (0.0, float(W), 0.0, float(H)), p0, sig2, isig2, right_u=ru_full)
p = ctx.profile_read()
print("fused  :", {k: round(ms / c, 4) for k, (ms, c) in p.items() if c})
for _ in range(20):
    ctx.pose_only_optimize(X[qi][g["assigned"][ef]].astype(np.float64), meas, isig2[oc].astype(np.float64), sig2[oc], p0, FX, FX, CX, CY, BF)
p = ctx.profile_read()
print("direct :", {k: round(ms / c, 4) for k, (ms, c) in p.items() if c})
