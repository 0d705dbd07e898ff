#!/bin/bash
# Runs on the GPU box: the stamps build of the library (tools/exp/ab_build.sh qtst "-DQT_STAMPS") for a few single-pair extractions; the
# seventh level-0 tree prints its timeline (cycles per section / batched step)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
cp orb_slam2_ros2_amd/liborbfe_hip.so /tmp/liborbfe_keep.so
cp tools/exp/libs/liborbfe_qtst.so orb_slam2_ros2_amd/liborbfe_hip.so
python3 - <<'PY' 2>&1 | tail -60
import sys; sys.path.insert(0, ".")
import numpy as np
from orb_slam2_ros2_amd import synth
from orb_slam2_ros2_amd._lib import Context
L, R = synth.stereo_pair(0)
ctx = Context(1241, 376, max_images=2)
import os
if os.environ.get("QT_BATCH"):
    import torch
    n = int(os.environ["QT_BATCH"])
    ctx.close()
    ctx = Context(1241, 376, max_images=2 * n)
    dl = torch.from_numpy(np.stack([L] * n)).cuda()
    dr = torch.from_numpy(np.stack([R] * n)).cuda()
    ctx.stereo_batch_device(dl.data_ptr(), dr.data_ptr(), 1241, 1241 * 376, n, 718.856, 386.1)
    ctx.sync()
else:
    for _ in range(6):
        ctx.extract_batch([L, R])
    ctx.sync()
ctx.close()
PY
cp /tmp/liborbfe_keep.so orb_slam2_ros2_amd/liborbfe_hip.so
