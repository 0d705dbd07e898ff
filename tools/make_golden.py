#!/usr/bin/env python3
"""Generate tests/golden/golden_v1.json from the CPU oracle on the seeded synthetic inputs of SURVEY.md 8(d).

The reference itself cannot run here (no OpenCV/g2o), so these vectors pin the ORACLE (regression + GPU parity
target), not the reference: "parity unpinned" in the sense of the task statement.  Re-run after any intended
change of the oracle or of the synthetic generator and commit the result."""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle.pyoracle import Oracle
from orb_slam2_ros2_amd import ba_synth, synth
from orb_slam2_ros2_amd.digest import pair_digest


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    orc = Oracle()
    g = {"version": 1, "frames": {}, "blur_variant": 0}
    for f in (0, 1, 7):
        L, R = synth.stereo_pair(f)
        r = orc.stereo_frame(L, R, fx=718.856, bf=718.856 * 0.537166, math_mode=0, threads=1)
        g["frames"][f"kitti_{f}"] = {
            "left_sha": sha(L), "right_sha": sha(R), "n_left": len(r["lk"]), "n_right": len(r["rk"]), "n_matches": int(r["n_matches"]),
            "lk_sha": sha(r["lk"]), "ld_sha": sha(r["ld"]), "rk_sha": sha(r["rk"]), "rd_sha": sha(r["rd"]),
            "right_u_sha": sha(r["right_u"]), "depth_sha": sha(r["depth"]),
            "per_level_left": [int((r["lk"]["octave"] == l).sum()) for l in range(8)],
            "first_kps": [[float(k[n]) for n in ("x", "y", "angle", "response")] + [int(k["octave"])] for k in r["lk"][:6]],
            "first_desc": r["ld"][:2].tolist(),
        }
    # the frames bench.py runs (rank r: frames 16 r ... 16 r + 15, up to 8 ranks): one digest per pair, compared after the timed region
    g["bench_pairs"] = {}
    for f in range(128):
        L, R = synth.stereo_pair(f)
        r = orc.stereo_frame(L, R, fx=718.856, bf=718.856 * 0.537166, math_mode=0, threads=2)
        g["bench_pairs"][str(f)] = pair_digest(r["lk"], r["ld"], r["rk"], r["rd"], r["right_u"], r["depth"], r["n_matches"])
    img = synth.mono_image(0)
    ex = orc.extractor(img, n_features=1000)
    k, d = ex.extract()
    g["frames"]["tum_0"] = {"img_sha": sha(img), "n": len(k), "k_sha": sha(k), "d_sha": sha(d),
                            "level_dims": [list(ex.level_info(l)[:2]) for l in range(8)], "quotas": [ex.level_info(l)[3] for l in range(8)]}
    Ls, _ = synth.stereo_pair(5, sparse=True)
    ks, ds = orc.extractor(Ls).extract()
    g["frames"]["sparse_5"] = {"img_sha": sha(Ls), "n": len(ks), "k_sha": sha(ks), "per_level": [int((ks["octave"] == l).sum()) for l in range(8)]}
    q, t = synth.descriptors_cfg3()
    bi, bd, sd = orc.match_bruteforce(q, t)
    g["cfg3"] = {"q_sha": sha(q), "t_sha": sha(t), "best_idx_sha": sha(bi), "best_dist_sha": sha(bd), "second_sha": sha(sd),
                 "n_second_intmax": int((sd == 2**31 - 1).sum()), "best_dist_hist": np.bincount(np.minimum(bd, 255) // 32, minlength=8).tolist()}
    p = ba_synth.make_problem()
    o = orc.ba_eval_edges(p["poses"], p["points"], p["edge_pose"], p["edge_point"], p["meas"], p["is_stereo"], p["info"],
                          p["huber_delta"], p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
    g["cfg5_ba"] = {"n_edges": int(p["edge_pose"].size), "n_stereo": int(p["is_stereo"].sum()), "meas_sha": sha(p["meas"]),
                    "chi2_sum": float(o["chi2"].sum()), "rho_sum": float(o["rho"][:, 0].sum()), "chi2_sha": sha(o["chi2"]),
                    "jpose_abs_sum": float(np.abs(o["j_pose"]).sum())}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "golden_v1.json")
    with open(out, "w") as fh:
        json.dump(g, fh, indent=1, sort_keys=True)
    print("wrote", out)


if __name__ == "__main__":
    main()
