#!/usr/bin/env python3
"""Generate tests/golden/golden_v2.json: oracle outputs for the widened rows (local BA, pose-only, frame glue) on seeded synthetic
inputs.  Like golden_v1 these pin the ORACLE (parity unpinned by the reference).  Floating-point results are stored to 12 significant
digits and compared with a tolerance."""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle.pyoracle import Oracle
from orb_slam2_ros2_amd import ba_synth, synth

TUM = dict(fx=520.908620, fy=521.007327, cx=325.141442, cy=249.701764, k1=0.231222, k2=-0.784899, p1=-0.003257, p2=-0.000105, k3=0.917205, bf=40.0)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def color_image(seed, w=640, h=480):
    g = synth.mono_image(seed, w, h)
    rng = np.random.default_rng(seed)
    img = np.stack([g, np.roll(g, 3, 1), (255 - g)], 2).astype(np.int32) + rng.integers(-6, 7, (h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def r12(a):
    return [float(f"{x:.12g}") for x in np.asarray(a, np.float64).ravel()]


def main():
    orc = Oracle()
    g = {"version": 2}
    pr = ba_synth.make_problem(seed=3, n_kf=12, n_pt=400, with_truth=True)
    fixed = np.zeros(12, np.uint8)
    fixed[:2] = 1
    pr["poses"][:2] = pr["poses_true"][:2]
    r = orc.ba_local_optimize(pr, fixed)
    g["lba"] = {"iters": r["iters"].tolist(), "poses": r12(r["poses"]), "points_head": r12(r["points"][:20]), "chi2_sum": float(f"{r['chi2'].sum():.12g}"),
                "n_level1": int(r["level"].sum()), "n_bad": int(r["bad"].sum()), "level_sha": sha(r["level"]), "bad_sha": sha(r["bad"])}
    p = ba_synth.make_pose_problem()
    n_good, pose, inl = orc.pose_only_optimize(p["Xw"], p["meas"], p["info"], p["sigma2"], p["pose"], p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
    g["pose_only"] = {"n_good": int(n_good), "pose": r12(pose), "inlier_sha": sha(inl.astype(np.uint8))}
    img = color_image(3)
    g["glue"] = {"img_sha": sha(img), "gray_rgb_sha": sha(orc.cvt_gray(img, 1)), "gray_bgr_sha": sha(orc.cvt_gray(img, 2))}
    K = np.array([TUM[q] for q in ("fx", "fy", "cx", "cy")], np.float32)
    D = np.array([TUM[q] for q in ("k1", "k2", "p1", "p2", "k3")], np.float32)
    pts = np.stack([np.linspace(30, 610, 40), np.linspace(25, 455, 40)[::-1]], 1).astype(np.float32)
    und = orc.undistort_points(pts, K, D)
    g["glue"]["undistort_sha"] = sha(und)
    g["glue"]["undistort_head"] = [[float(x) for x in row] for row in und[:4]]
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "golden_v2.json")
    with open(out, "w") as fh:
        json.dump(g, fh, indent=1, sort_keys=True)
    print("wrote", out)


if __name__ == "__main__":
    main()
