#!/bin/bash
# Runs on the GPU box: tools/latency_single.py with library variants (tools/exp/libs/liborbfe_<v>.so), alternating
cd ${GRAFT_REPO_ROOT:-.}
cp orb_slam2_ros2_amd/liborbfe_hip.so /tmp/keep.so
for round in 1 2 3; do
  for v in $1; do
    cp tools/exp/libs/liborbfe_$v.so orb_slam2_ros2_amd/liborbfe_hip.so
    echo -n "$v: "; timeout -k 10 120 python3 tools/latency_single.py 2>&1 | grep -v amdgpu.ids | head -2 | tr '\n' ' '; echo
  done
done
cp /tmp/keep.so orb_slam2_ros2_amd/liborbfe_hip.so
