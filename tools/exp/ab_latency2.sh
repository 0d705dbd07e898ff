#!/bin/bash
# Runs on the GPU box: tools/exp/ab_latency2.sh "<variants>" -- single-pair latency (tools/latency_single.py) of each variant library, alternating, two rounds
R=$GRAFT_REPO_ROOT
cp $R/orb_slam2_ros2_amd/liborbfe_hip.so /tmp/keep.so
for round in 1 2; do
  for v in $1; do
    cp $R/tools/exp/libs/liborbfe_$v.so $R/orb_slam2_ros2_amd/liborbfe_hip.so
    echo "== $v"; timeout -k 10 200 python3 $R/tools/latency_single.py 2>&1 | head -2 | cut -c1-400
  done
done
cp /tmp/keep.so $R/orb_slam2_ros2_amd/liborbfe_hip.so
