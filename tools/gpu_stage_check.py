#!/usr/bin/env python3
"""Stage-by-stage GPU-vs-oracle diagnostic (prints every mismatch class instead of stopping at the first).
Usage: python tools/gpu_stage_check.py [n_frames]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle.pyoracle import Oracle
from orb_slam2_ros2_amd import synth
from orb_slam2_ros2_amd._lib import Context

def main():
    nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    orc = Oracle()
    ctx = Context(1241, 376, max_images=2)
    fx, bf = 718.856, 386.1448
    bad = 0
    for f in range(nfr):
        L, R = synth.stereo_pair(f)
        t = time.time()
        (lk, ld), (rk, rd) = ctx.extract_batch([L, R])
        nm, ru, dp, br, bd = ctx.stereo_match(0, 1, fx, bf)
        print(f"frame {f}: gpu extract+match {1e3*(time.time()-t):.1f} ms  nL={len(lk)} nR={len(rk)} matches={nm}")
        for slot, img, gk, gd in ((0, L, lk, ld), (1, R, rk, rd)):
            ex = orc.extractor(img, math_mode=1)
            ok, od = ex.extract()
            for l in range(8):
                for blurred in (False, True):
                    a, b = ctx.pyramid(slot, l, blurred), ex.plane(l, blurred)
                    if not np.array_equal(a, b):
                        d = np.argwhere(a != b)
                        print(f"  MISMATCH slot{slot} level{l} blurred={blurred}: {len(d)} px, first {d[0]} gpu={a[tuple(d[0])]} cpu={b[tuple(d[0])]}")
                        bad += 1
                gc, oc = ctx.debug_candidates(slot, l), ex.candidates(l)
                if gc.shape != oc.shape or not np.array_equal(gc, oc):
                    print(f"  MISMATCH slot{slot} level{l} candidates: gpu {gc.shape} cpu {oc.shape}")
                    n = min(len(gc), len(oc))
                    dif = np.argwhere((gc[:n] != oc[:n]).any(axis=1))
                    if len(dif): print("    first diff at", dif[0], gc[dif[0][0]], oc[dif[0][0]])
                    bad += 1
            if len(gk) != len(ok):
                print(f"  MISMATCH slot{slot} n keypoints gpu {len(gk)} cpu {len(ok)}")
                for l in range(8): print("    level", l, (gk['octave']==l).sum(), (ok['octave']==l).sum())
                bad += 1
            else:
                for fld in ("x", "y", "size", "angle", "response", "octave", "class_id"):
                    ne = np.argwhere(gk[fld] != ok[fld])
                    if len(ne):
                        i = ne[0][0]
                        print(f"  MISMATCH slot{slot} kp.{fld}: {len(ne)} differ, first i={i} gpu={gk[fld][i]!r} cpu={ok[fld][i]!r}")
                        bad += 1
                nd = np.argwhere((gd != od).any(axis=1))
                if len(nd):
                    print(f"  MISMATCH slot{slot} descriptors: {len(nd)} rows differ, first {nd[0][0]}")
                    bad += 1
            # libm-mode oracle must agree with the deterministic-math oracle on the final outputs
            ex0 = orc.extractor(img, math_mode=0)
            k0, d0 = ex0.extract()
            if not (np.array_equal(k0, ok) and np.array_equal(d0, od)):
                print(f"  NOTE slot{slot}: libm oracle differs from det-math oracle")
                bad += 1
            if slot == 0: exl, okl, odl = ex, ok, od
            else: exr, okr, odr = ex, ok, od
        om, oru, odp, obr, obd = exl.stereo_match(exr, okl, odl, okr, odr, fx, bf)
        n = len(okl)
        for name, a, b in (("right_u", ru[:n], oru), ("depth", dp[:n], odp), ("best_right", br[:n], obr), ("best_dist", bd[:n], obd)):
            ne = np.argwhere(a != b)
            if len(ne):
                i = ne[0][0]
                print(f"  MISMATCH stereo {name}: {len(ne)} differ, first i={i} gpu={a[i]!r} cpu={b[i]!r}")
                bad += 1
        if om != nm:
            print(f"  MISMATCH stereo n_matches gpu {nm} cpu {om}"); bad += 1
    print("RESULT:", "ALL STAGES BIT-EXACT" if bad == 0 else f"{bad} mismatch classes")
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
