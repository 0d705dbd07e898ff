#!/usr/bin/env python3
"""Per content class (synth.CONTENT_CLASSES): what the ORACLE sees on a few frames -- FAST candidates and selected keypoints per level,
the share of cells that have no corner at the high threshold (they repeat at the low one, ORBExtractor.cc:365-367), the levels that
return nothing (quirk Q3), stereo matches.  CPU only; used to shape the generators, numbers quoted in DESIGN.md."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle.pyoracle import Oracle
from orb_slam2_ros2_amd import synth


lo_pass_cells = synth.lo_pass_cells


def main():
    orc = Oracle()
    for cls in synth.CONTENT_CLASSES:
        for f in (0, 1):
            L, R = synth.stereo_pair_content(f, cls)
            ex = orc.extractor(L)
            k, d = ex.extract()
            r = orc.stereo_frame(L, R, math_mode=0, threads=2)
            nc, ns, lo, cells = [], [], 0, 0
            for l in range(8):
                c = ex.candidates(l)
                wl, hl = ex.level_info(l)[:2]
                a, b = lo_pass_cells(c, wl - 32, hl - 32)
                lo += a
                cells += b
                nc.append(len(c))
                ns.append(int((k["octave"] == l).sum()))
            print(f"{cls:10s} f{f}: cand {sum(nc):6d} {nc}  sel {len(k):5d} {ns}  lo-pass cells {lo}/{cells} = {lo / cells:.2f}  matches {r['n_matches']}"
                  f"  mean {L.mean():.0f} std {L.std():.0f}")


if __name__ == "__main__":
    main()
