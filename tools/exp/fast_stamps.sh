#!/bin/bash
# Runs on the GPU box: the FAST_STAMPS build (tools/exp/ab_build.sh fst "-DFAST_STAMPS") on single pairs -- where a one-cell wave's life goes
cd ${GRAFT_REPO_ROOT:-.}
cp orb_slam2_ros2_amd/liborbfe_hip.so /tmp/keep.so
cp tools/exp/libs/liborbfe_fst.so orb_slam2_ros2_amd/liborbfe_hip.so
python3 - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, ctypes; sys.path.insert(0, ".")
from orb_slam2_ros2_amd import synth
from orb_slam2_ros2_amd._lib import Context, load
L, R = synth.stereo_pair(0)
lib = load()
ctx = Context(1241, 376, max_images=2)
for _ in range(5): ctx.extract_batch([L, R])
ctx.sync()
lib.orbfe_debug_fast_stamps()       # (warm-up: reset)
for _ in range(20): ctx.extract_batch([L, R])
ctx.sync()
lib.orbfe_debug_fast_stamps()
ctx.close()
PY
cp /tmp/keep.so orb_slam2_ros2_amd/liborbfe_hip.so
