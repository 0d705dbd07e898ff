#!/bin/bash
# Runs on the GPU box: tools/exp/ab_run.sh "<variants>" [pairs] [steps] -- wall-clock step time of each variant library, alternating, three rounds
R=$GRAFT_REPO_ROOT
cp $R/orb_slam2_ros2_amd/liborbfe_hip.so /tmp/keep.so
for round in 1 2 3; do
  for v in $1; do
    cp $R/tools/exp/libs/liborbfe_$v.so $R/orb_slam2_ros2_amd/liborbfe_hip.so
    echo -n "$v: "; timeout -k 10 200 python3 $R/tools/step_time.py ${2:-512} ${3:-150} 2>&1 | tail -1
  done
done
cp /tmp/keep.so $R/orb_slam2_ros2_amd/liborbfe_hip.so
