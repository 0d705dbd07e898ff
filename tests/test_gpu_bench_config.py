"""GPU parity of the configuration bench.py times: orbfe_stereo_batch_device at 512 pairs = 1024 images per
call -- at that size the quadtree keeps its candidate records in global memory (32 trees per CU leave no LDS for them), a path the
smaller tests never take -- and of every ORBFE_* runtime switch, each in a fresh process."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from orb_slam2_ros2_amd import synth
from orb_slam2_ros2_amd.digest import batch_digests

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_v1.json")))
FX, BF = 718.856, 718.856 * 0.537166


def test_bench_batch_of_512_pairs_equals_the_oracle(orc):
    """Every one of the 512 pairs of a bench step (16 distinct frames tiled 32 times, as bench.py builds its batch) against the oracle's
    result for its frame: keypoints, descriptors, right_u, depth, match count -- arrays for the distinct frames, digests for all."""
    import torch
    from orb_slam2_ros2_amd._lib import Context
    B, U = 512, 16
    frames = [synth.stereo_pair(f) for f in range(U)]
    ref = [orc.stereo_frame(L, R, fx=FX, bf=BF) for L, R in frames]
    ctx = Context(1241, 376, max_images=2 * B)
    dl = torch.from_numpy(np.stack([frames[i % U][0] for i in range(B)])).cuda()
    dr = torch.from_numpy(np.stack([frames[i % U][1] for i in range(B)])).cuda()
    for _ in range(3):   # back to back like the timed loop: pipelined stereo match, both pyramid buffers
        ctx.stereo_batch_device(dl.data_ptr(), dr.data_ptr(), 1241, 1241 * 376, B, FX, BF)
    ctx.sync()
    kps, desc, cnt = ctx.fetch_batch(0, 2 * B)
    ru, dp, nm = ctx.fetch_stereo_batch(0, B)
    for p in range(B):
        r = ref[p % U]
        nl, nr = len(r["lk"]), len(r["rk"])
        assert (cnt[2 * p], cnt[2 * p + 1], nm[p]) == (nl, nr, r["n_matches"]), f"pair {p}: counts"
        assert np.array_equal(kps[2 * p, :nl], r["lk"]) and np.array_equal(desc[2 * p, :nl], r["ld"]), f"pair {p}: left features"
        assert np.array_equal(kps[2 * p + 1, :nr], r["rk"]) and np.array_equal(desc[2 * p + 1, :nr], r["rd"]), f"pair {p}: right features"
        assert np.array_equal(ru[p, :nl].view(np.int64), r["right_u"].view(np.int64)), f"pair {p}: right_u"
        assert np.array_equal(dp[p, :nl].view(np.int64), r["depth"].view(np.int64)), f"pair {p}: depth"
    dig = batch_digests(kps, desc, cnt, ru, dp, nm)
    assert all(dig[p] == G["bench_pairs"][str(p % U)] for p in range(B))
    # the per-level FAST candidate sets of a slot deep in the batch (global-record quadtree input), frame 13
    ex = orc.extractor(frames[13][0])
    ex.extract()
    slot = 2 * (13 + 16 * 20)
    for l in range(8):
        assert np.array_equal(ctx.debug_candidates(slot, l), ex.candidates(l)), f"level {l} candidates of slot {slot}"
    ctx.close()


KNOBS = [   # the nine switches the library still reads (orbfe_create), each a fresh process
    {},                                   # the production schedule
    {"ORBFE_QT_REC_CAP": "0"},            # quadtree: records in global memory + bounce buffer (what 1024 images per launch use)
    {"ORBFE_QT_REC_CAP": "600"},          # ... and a partial LDS cache
    {"ORBFE_QT_LDS_NODES": "300"},        # node tables of the levels with quota > ~290 in GLOBAL memory (what nFeatures > ~12 000 uses)
    {"ORBFE_LBA_HOST_LM": "1"},           # (no effect on this path; the switch must at least not break context creation)
    {"ORBFE_PIPELINE_STEREO": "0"},       # stereo match in line
    {"ORBFE_OVERLAP_BLUR": "0"},          # blur in line, no second stream
    {"ORBFE_FAST_CPW": "4"},              # k_fast: four cells per wave in every launch (default: only where a launch holds >= 65536 cells)
    {"ORBFE_FAST_CPW": "7"},
    {"ORBFE_NO_XCD_ORDER": "1"},          # row-major cell / tile tables
    {"ORBFE_GRAPHS": "0"},                # no hipGraph replay on the host-pointer path
    {"ORBFE_QT_REC_CAP": "0", "ORBFE_PIPELINE_STEREO": "0", "ORBFE_OVERLAP_BLUR": "0", "ORBFE_GRAPHS": "0"},
]


@pytest.mark.parametrize("knob", KNOBS, ids=lambda k: ",".join(f"{a}={b}" for a, b in k.items()) or "default")
def test_runtime_switches_do_not_change_results(knob):
    env = dict(os.environ)
    for k in list(env):
        if k.startswith("ORBFE_"):
            del env[k]
    env.update(knob)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "knob_check.py"), "32"], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and "KNOB_OK" in out.stdout, f"{knob}: rc {out.returncode}\n{out.stdout[-2000:]}\n{out.stderr[-4000:]}"
