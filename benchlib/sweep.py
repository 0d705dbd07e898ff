"""bench.py's content sweep: the headline step on every synthetic content class, every pair verified against the oracle's digests."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

from .common import *  # noqa: F401,F403
from .common import _cpu_ms, _kernel_us, _oracle_fast, _sha, _stats_ms  # noqa: F401


def content_sweep(ctx, B, dev, rect_hosts, steps, rect_want):
    """The same step on every content class of synth.CONTENT_CLASSES: the reference's input contract is a camera
    image (example/Stereo/KittiStereo.cc:28-33) and the cost of the path depends on the content -- cells that repeat cv::FAST at the
    low threshold (ORBExtractor.cc:365-367), candidates the quadtree spreads, right keypoints per row band.  Per class: 16 distinct
    pairs tiled to B ("rect": the headline batch itself, B distinct pairs, digests rect_want), device-resident; stage times with every kernel ALONE (HIP events), then `steps` steps of
    the production schedule; EVERY pair of the last batch checked against the committed oracle digests (golden_v1 bench_pairs for
    "rect", golden_v4 for the rest, tools/make_golden_v4.py) before a number is reported."""
    import torch

    from orb_slam2_ros2_amd import synth
    from orb_slam2_ros2_amd.digest import batch_digests
    g4 = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_v4.json")))["classes"]
    U = min(B, 16)
    out = {}
    for cls in synth.CONTENT_CLASSES:
        if cls == "rect":
            left_h, right_h = rect_hosts
            gold = None
        else:
            fr = [synth.stereo_pair_content(f, cls, W, H) for f in range(U)]
            reps = (B + U - 1) // U
            left_h = np.stack(([a for a, _ in fr] * reps)[:B])
            right_h = np.stack(([b for _, b in fr] * reps)[:B])
            gold = g4[cls]["pairs"]
        dl, dr = torch.from_numpy(left_h).to(dev), torch.from_numpy(right_h).to(dev)

        def step():
            ctx.stereo_batch_device(dl.data_ptr(), dr.data_ptr(), W, W * H, B, FX, BF)
        for _ in range(5):
            step()
        ctx.sync()
        ctx.profile_enable(1)
        ctx.profile_read()
        for _ in range(5):
            step()
        ctx.sync()
        alone = {k: ms / n for k, (ms, n) in ctx.profile_read().items() if n}
        ctx.profile_enable(0)
        for _ in range(3):
            step()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ctx.sync()
        ms_step = (time.perf_counter() - t0) / steps * 1e3
        kps, desc, cnt = ctx.fetch_batch(0, 2 * B)
        ru, dp, nm = ctx.fetch_stereo_batch(0, B)
        dig = batch_digests(kps, desc, cnt, ru, dp, nm)
        if cls == "rect":
            bad = [p_ for p_ in range(B) if rect_want[p_] is not None and dig[p_][:len(rect_want[p_])] != rect_want[p_]]
        else:
            bad = [p_ for p_ in range(B) if dig[p_] != gold[str(p_ % U)]]
        if bad:
            raise SystemExit(f"bench.py: content sweep: class {cls}: {len(bad)} of {B} pairs differ from the golden digests (first: pair {bad[0]})")
        lo = cells = n_cand = 0
        for l in range(NLEVELS):
            c = ctx.debug_candidates(0, l)
            li = ctx.level_info(l)
            a, b = synth.lo_pass_cells(c, li.width - 32, li.height - 32, TH_HI)
            lo, cells, n_cand = lo + a, cells + b, n_cand + len(c)
        out[cls] = {"ms_per_step": ms_step, "pairs_per_s": B / ms_step * 1e3, "fast_ms": alone.get("fast"), "quadtree_ms": alone.get("quadtree"),
                    "stereo_ms": alone.get("stereo"), "resize_ms": alone.get("resize"), "blur_ms": alone.get("blur"),
                    "orient_brief_ms": alone.get("orient_brief"), "frac_cells_lo_pass": lo / max(cells, 1), "candidates_per_image": n_cand,
                    "keypoints_per_image": float(cnt.mean()), "matches_per_pair": float(nm.mean()), "verified_pairs": B, "steps": steps,
                    "distinct_pairs": len(set(rect_want)) if cls == "rect" else U}
        del dl, dr, kps, desc, ru, dp
        torch.cuda.empty_cache()
    ms = [v["ms_per_step"] for v in out.values()]
    out["worst_over_best"] = max(ms) / min(ms)
    out["what"] = ("the headline step (512 pairs resident in HBM, production schedule) per synthetic content class, every pair verified against the "
                   "oracle's digests; *_ms: the stage's kernels ALONE (HIP events, untimed pass); frac_cells_lo_pass / candidates_per_image: the left "
                   "image of frame 0 (cells without a corner at 20 repeat cv::FAST at 7); 'rect' is the class `value` is quoted on")
    return out
