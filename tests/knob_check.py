#!/usr/bin/env python3
"""Run by tests/test_gpu_bench_config.py in a FRESH process per setting (the ORBFE_* switches are read at orbfe_create): the device
batch path, the host-pointer path and the per-slot path on the frames bench.py uses, compared with the committed digests of
tests/golden/golden_v1.json (made by the oracle; no oracle is needed here).  Prints KNOB_OK on success."""
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FX, BF = 718.856, 718.856 * 0.537166


def main():
    import torch

    from orb_slam2_ros2_amd import synth
    from orb_slam2_ros2_amd._lib import Context
    from orb_slam2_ros2_amd.digest import batch_digests, pair_digest

    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_v1.json")))["bench_pairs"]
    frames = [synth.stereo_pair(f) for f in range(16)]
    ctx = Context(1241, 376, max_images=2 * n_pairs)
    dl = torch.from_numpy(np.stack([frames[i % 16][0] for i in range(n_pairs)])).cuda()
    dr = torch.from_numpy(np.stack([frames[i % 16][1] for i in range(n_pairs)])).cuda()
    for rep in range(3):  # back to back: the pipelined schedule, both pyramid buffers in both roles
        ctx.stereo_batch_device(dl.data_ptr(), dr.data_ptr(), 1241, 1241 * 376, n_pairs, FX, BF)
    ctx.sync()
    kps, desc, cnt = ctx.fetch_batch(0, 2 * n_pairs)
    ru, dp, nm = ctx.fetch_stereo_batch(0, n_pairs)
    dig = batch_digests(kps, desc, cnt, ru, dp, nm)
    bad = [p for p in range(n_pairs) if dig[p] != gold[str(p % 16)]]
    assert not bad, f"device batch: pairs {bad[:8]} ... differ from the golden digests ({len(bad)} of {n_pairs})"

    # host-pointer path (hipGraph replay unless ORBFE_GRAPHS=0), twice so the replay is used
    for f in (3, 3, 5):
        (lk, ld), (rk, rd) = ctx.extract_batch(list(frames[f]))
        m, r, d, _, _ = ctx.stereo_match(0, 1, FX, BF)
        assert pair_digest(lk, ld, rk, rd, r, d, m) == gold[str(f)], f"host path, frame {f}"

    # per-slot path on two threads (Frame.cc:100-105), slots 2 / 3
    for f in (7, 7, 9):
        out = {}

        def run(slot, img):
            out[slot] = ctx.extract_slot(slot, img)
        th = [threading.Thread(target=run, args=(2 + s, frames[f][s])) for s in (0, 1)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        m, r, d, _, _ = ctx.stereo_match(2, 3, FX, BF)
        assert pair_digest(out[2][0], out[2][1], out[3][0], out[3][1], r, d, m) == gold[str(f)], f"slot path, frame {f}"
    # ... and both eyes as one launch sequence on a slot lane (orbfe_extract_slots), twice so the captured graph is replayed
    for f in (11, 11, 4):
        (lk, ld), (rk, rd) = ctx.extract_slots(2, list(frames[f]))
        m, r, d, _, _ = ctx.stereo_match(2, 3, FX, BF)
        assert pair_digest(lk, ld, rk, rd, r, d, m) == gold[str(f)], f"extract_slots, frame {f}"
    # ... and the whole frame as one call (orbfe_frame_stereo / _slots): the match inside the extraction's launch sequence
    for f, slot in ((6, None), (6, None), (12, 4), (12, 4), (2, None)):
        (lk, ld), (rk, rd), m, r, d = ctx.frame_stereo(*frames[f], FX, BF, slot_left=slot)
        assert pair_digest(lk, ld, rk, rd, r, d, m) == gold[str(f)], f"frame_stereo, frame {f}, slot {slot}"
    # ... and an RGB-D frame as one call (orbfe_frame_rgbd_image) against the two calls it stands for, on two other slots
    cam = dict(fx=718.856, fy=718.856, cx=607.19, cy=185.22, k1=0.05, k2=-0.02, p1=0.001, p2=-0.0005, k3=0.01, bf=386.14)
    dep = (np.arange(376 * 1241, dtype=np.uint32).reshape(376, 1241) * 2654435761 >> 17).astype(np.uint16)
    for f in (8, 8, 13):
        ku, d, dd, ru = ctx.frame_rgbd_image(frames[f][0], cam, dep, 5000.0, 0, slot=6)
        k0, d0 = ctx.extract_slot(7, frames[f][0])
        ku0, dd0, ru0 = ctx.frame_rgbd(7, cam, dep, 5000.0)
        assert len(ku) == len(k0) and np.array_equal(d, d0) and ku.tobytes() == ku0[:len(ku)].tobytes(), f"frame_rgbd_image, frame {f}"
        assert np.array_equal(dd, dd0) and np.array_equal(ru, ru0), f"frame_rgbd_image depth, frame {f}"
    ctx.close()
    print("KNOB_OK", n_pairs)


if __name__ == "__main__":
    main()
