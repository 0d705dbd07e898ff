#!/bin/bash
# Runs on the GPU box: the BRIEF_STAMPS build (tools/exp/ab_build.sh bst "-DBRIEF_STAMPS") on single pairs, several processes (the stage is bimodal per process)
cd ${GRAFT_REPO_ROOT:-.}
cp orb_slam2_ros2_amd/liborbfe_hip.so /tmp/keep.so
cp tools/exp/libs/liborbfe_bst.so orb_slam2_ros2_amd/liborbfe_hip.so
for run in 1 2 3 4 5 6; do
python3 - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0, ".")
from orb_slam2_ros2_amd import synth
from orb_slam2_ros2_amd._lib import Context, load
L, R = synth.stereo_pair(0)
lib = load()
ctx = Context(1241, 376, max_images=2)
for _ in range(20): ctx.extract_batch([L, R])
ctx.sync()
lib.orbfe_debug_brief_stamps()
ctx.close()
PY
done
cp /tmp/keep.so orb_slam2_ros2_amd/liborbfe_hip.so
