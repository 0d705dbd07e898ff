#!/usr/bin/env python3
"""Wall-clock of the other BASELINE.json configurations through the host-pointer C-ABI (host buffers in, host results out, so
PCIe both ways is INCLUDED), with the CPU oracle timed beside each on one host core.  These are parity-test cases, not bench lines
(bench.py measures config 1/2 batched); the figures go into DESIGN.md section 5.  Usage: secondary_times.py [out.json]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_ros2_amd import ba_synth, synth
from orb_slam2_ros2_amd._lib import Context
from oracle import pyoracle

orc = pyoracle.Oracle(pyoracle.build(fast=True, out_dir=os.path.join("/tmp", f"orb_oracle_{os.getuid()}")))


def med(f, n=20, warm=3):
    for _ in range(warm):
        f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


out = {}
# config 3: 2000 x 2000 Hamming-256 brute force
ctx = Context(1241, 376, max_images=2)
q, t = synth.descriptors_cfg3()
g = med(lambda: ctx.match_bruteforce(q, t))
cand = np.arange(t.shape[0], dtype=np.uint32)
t0 = time.perf_counter()
for i in range(200):
    orc.best_match(q[i], t, cand)
c = (time.perf_counter() - t0) / 200 * q.shape[0] * 1e3
out["cfg3_bruteforce_2000x2000"] = {"gpu_ms": g, "cpu_oracle_ms": c, "gpu_Gpair_per_s": q.shape[0] * t.shape[0] / (g * 1e-3) / 1e9,
                                    "algorithmic_bytes": 2 * 2000 * 32 + 2000 * 12}
# config 2: one KITTI frame, host image -> host keypoints + descriptors
l, r = synth.stereo_pair(0)
g = med(lambda: ctx.extract(l))
c = med(lambda: orc.extractor(l).extract(), n=5, warm=1)
out["cfg2_single_kitti_frame_2000"] = {"gpu_ms": g, "cpu_oracle_ms": c}
ctx.close()
# config 5: TUM frame with 1000 features, BA edge evaluation / normal equations / local BA on the synthetic local map
ctx = Context(640, 480, n_features=1000, max_images=1)
img = synth.mono_image(0)
g = med(lambda: ctx.extract(img))
c = med(lambda: orc.extractor(img, n_features=1000).extract(), n=5, warm=1)
out["cfg5_single_tum_frame_1000"] = {"gpu_ms": g, "cpu_oracle_ms": c}
p = ba_synth.make_problem()
n_e = len(p["edge_pose"])
g = med(lambda: ctx.ba_eval_edges(**p))
gl = med(lambda: ctx.ba_eval_edges(**p, jacobians=False))
c = med(lambda: orc.ba_eval_edges(**p), n=5, warm=1)
out["cfg5_ba_edge_eval"] = {"edges": n_e, "gpu_ms": g, "gpu_ms_without_jacobians": gl, "cpu_oracle_ms": c,
                            "algorithmic_bytes": 304 * n_e + 60 * 56 + 3000 * 24}
fixed = np.zeros(p["poses"].shape[0], np.uint8)
fixed[0] = 1
fixed[30:] = 1
g = med(lambda: ctx.ba_build_system(**p, pose_fixed=fixed))
c = med(lambda: orc.ba_build_system(**p, pose_fixed=fixed), n=5, warm=1)
out["cfg5_ba_normal_equations"] = {"gpu_ms": g, "cpu_oracle_ms": c}
pr = ba_synth.make_problem(seed=42, n_kf=60, n_pt=3000, with_truth=True)
fx = np.zeros(60, np.uint8)
fx[:20] = 1
pr["poses"][:20] = pr["poses_true"][:20]
g = med(lambda: ctx.ba_local_optimize(pr, fx), n=5, warm=1)
c = med(lambda: orc.ba_local_optimize(pr, fx), n=3, warm=1)
out["cfg5_local_ba_5_plus_10_iterations"] = {"gpu_ms": g, "cpu_oracle_ms": c}
ctx.close()
for k, v in out.items():
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items()})
if len(sys.argv) > 1:
    json.dump({"note": "host-pointer C-ABI calls, median wall-clock incl. PCIe both ways; CPU = oracle/ on one host core; tools/secondary_times.py",
               "results": out}, open(sys.argv[1], "w"), indent=1, sort_keys=True)
