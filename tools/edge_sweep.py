#!/usr/bin/env python3
"""Ad-hoc sweep of extreme geometries: GPU vs oracle, bit-exact (used while developing; the pytest fuzz covers a random subset)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_ros2_amd import synth
from orb_slam2_ros2_amd._lib import Context, OrbfeError
from oracle import pyoracle
orc = pyoracle.Oracle(pyoracle.build())
cases = [(64, 64, 50, 1, 1.2), (97, 71, 200, 2, 1.5), (40, 200, 30, 1, 1.2), (2000, 120, 1500, 3, 1.2), (1241, 376, 1, 8, 1.2),
         (1241, 376, 7, 8, 1.2), (640, 480, 4000, 8, 1.2), (333, 333, 500, 8, 1.1), (800, 600, 2500, 4, 2.0), (1920, 1080, 5000, 8, 1.2)]
bad = 0
for w, h, nf, nl, sc in cases:
    img = synth.stereo_pair(3, w, h, n_rect=max(20, w * h // 2500))[0]
    try:
        ctx = Context(w, h, n_features=nf, n_levels=nl, scale_factor=sc, max_images=1)
    except OrbfeError as e:
        print((w, h, nf, nl, sc), "create refused:", str(e)[:90])
        continue
    k, d = ctx.extract(img)
    ok, od = orc.extractor(img, n_features=nf, n_levels=nl, scale=sc).extract()
    same = len(k) == len(ok) and np.array_equal(d, od) and all(np.array_equal(k[f].view(np.int32), ok[f].view(np.int32)) for f in ("x", "y", "angle", "octave", "response"))
    print((w, h, nf, nl, sc), len(k), "OK" if same else "MISMATCH")
    bad += not same
    ctx.close()
sys.exit(1 if bad else 0)
