#!/bin/bash
# Runs on the GPU box: the single-pair path with the small-launch quadtree compiled for 4 / 8 / 16 waves per tree (tools/exp/ab_build.sh qwN
# "-DQT_SMALL_WAVES=N"), alternating; parity of each variant against the oracle on the frame-or-two tests first
cd ${GRAFT_REPO_ROOT:-.}
cp orb_slam2_ros2_amd/liborbfe_hip.so /tmp/keep.so
for v in $1; do
  cp tools/exp/libs/liborbfe_$v.so orb_slam2_ros2_amd/liborbfe_hip.so
  echo "== $v parity"; timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
done
for round in 1 2 3; do
  for v in $1; do
    cp tools/exp/libs/liborbfe_$v.so orb_slam2_ros2_amd/liborbfe_hip.so
    echo -n "$v: "; timeout -k 10 120 python3 tools/latency_single.py 2>&1 | grep -v amdgpu.ids | head -2 | tr '\n' ' '; echo
  done
done
cp /tmp/keep.so orb_slam2_ros2_amd/liborbfe_hip.so
