#!/usr/bin/env python3
"""Generate tests/golden/golden_v4.json: the CPU oracle on the content classes of synth.stereo_pair_content.

Per class ("camera", "saturated", "sparse"; "rect" is golden_v1's bench_pairs) the digests of frames 0 .. 15 -- the frames bench.py's
content sweep tiles into its 512-pair batch -- and, for frames 0 and 1, what the throughput depends on: FAST candidates and selected
keypoints per level, the share of cells without a corner at the high threshold, stereo matches.  These vectors pin the ORACLE (the
reference cannot run here: "parity unpinned"); re-run after an intended change of the oracle or of the generator and commit the result."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from oracle.pyoracle import Oracle
from orb_slam2_ros2_amd import synth
from orb_slam2_ros2_amd.digest import pair_digest
from tools.content_stats import lo_pass_cells

FX, BF = 718.856, 718.856 * 0.537166


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    orc = Oracle()
    g = {"version": 4, "blur_variant": 0, "classes": {}}
    for cls in synth.CONTENT_CLASSES:
        if cls == "rect":
            continue
        ent = {"pairs": {}, "frames": {}}
        for f in range(16):
            L, R = synth.stereo_pair_content(f, cls)
            r = orc.stereo_frame(L, R, fx=FX, bf=BF, math_mode=0, threads=2)
            ent["pairs"][str(f)] = pair_digest(r["lk"], r["ld"], r["rk"], r["rd"], r["right_u"], r["depth"], r["n_matches"])
            if f < 2:
                ex = orc.extractor(L)
                k, _ = ex.extract()
                lo = cells = 0
                nc = []
                for l in range(8):
                    c = ex.candidates(l)
                    wl, hl = ex.level_info(l)[:2]
                    a, b = lo_pass_cells(c, wl - 32, hl - 32)
                    lo, cells = lo + a, cells + b
                    nc.append(len(c))
                ent["frames"][str(f)] = {"left_sha": sha(L), "right_sha": sha(R), "n_left": len(r["lk"]), "n_right": len(r["rk"]),
                                         "n_matches": int(r["n_matches"]), "candidates_per_level": nc,
                                         "selected_per_level": [int((k["octave"] == l).sum()) for l in range(8)],
                                         "cells": cells, "cells_lo_pass": lo}
        g["classes"][cls] = ent
        print(cls, {k: v for k, v in ent["frames"]["0"].items() if not k.endswith("_sha")})
    out = os.path.join(ROOT, "tests", "golden", "golden_v4.json")
    with open(out, "w") as fh:
        json.dump(g, fh, indent=1, sort_keys=True)
    print("wrote", out)


if __name__ == "__main__":
    main()
