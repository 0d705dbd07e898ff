#!/usr/bin/env python3
"""Generate tests/golden/golden_v3.json: oracle outputs on the BASELINE config-5 problem the bench's `ba` leg times (60 keyframes, 3000
points, ~15.6 k edges): the local BA (5 + 10 Levenberg-Marquardt iterations, 40 free keyframes) and the normal-equation
build.  Like golden_v1 / v2 these pin the ORACLE (parity unpinned by the reference).  Floats are stored to 12 significant digits and
compared with a tolerance."""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle.pyoracle import Oracle
from orb_slam2_ros2_amd import ba_synth


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def r12(a):
    return [float(f"{x:.12g}") for x in np.asarray(a, np.float64).ravel()]


def cfg5_local_ba_problem():
    """the problem tools/secondary_times.py and bench.py time: 20 fixed observers, 40 free keyframes"""
    pr = ba_synth.make_problem(seed=42, n_kf=60, n_pt=3000, with_truth=True)
    fixed = np.zeros(60, np.uint8)
    fixed[:20] = 1
    pr["poses"][:20] = pr["poses_true"][:20]
    return pr, fixed


def main():
    orc = Oracle()
    g = {"version": 3}
    pr, fixed = cfg5_local_ba_problem()
    r = orc.ba_local_optimize(pr, fixed)
    g["cfg5_lba"] = {"n_edges": int(pr["edge_pose"].size), "iters": r["iters"].tolist(), "poses": r12(r["poses"]),
                     "points_head": r12(r["points"][:20]), "chi2_sum": float(f"{r['chi2'].sum():.12g}"), "n_level1": int(r["level"].sum()),
                     "n_bad": int(r["bad"].sum())}
    p = ba_synth.make_problem()
    nk = p["poses"].shape[0]
    fx = np.zeros(nk, np.uint8)
    fx[0] = 1
    fx[30:] = 1
    s = orc.ba_build_system(**p, pose_fixed=fx)
    g["cfg5_system"] = {k: float(f"{np.abs(np.asarray(v, np.float64)).sum():.12g}") for k, v in s.items() if isinstance(v, np.ndarray) and v.dtype == np.float64}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "golden_v3.json")
    with open(out, "w") as fh:
        json.dump(g, fh, indent=1, sort_keys=True)
    print("wrote", out, {k: (v if not isinstance(v, list) else len(v)) for k, v in g["cfg5_lba"].items()}, g["cfg5_system"])


if __name__ == "__main__":
    main()
