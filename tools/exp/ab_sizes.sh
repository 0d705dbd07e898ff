#!/bin/bash
# Runs on the GPU box: tools/exp/ab_sizes.sh "<variants>" "<pairs list>" -- step time of each variant library at several batch sizes
R=$GRAFT_REPO_ROOT
cp $R/orb_slam2_ros2_amd/liborbfe_hip.so /tmp/keep.so
for n in $2; do
  for round in 1 2; do
    for v in $1; do
      cp $R/tools/exp/libs/liborbfe_$v.so $R/orb_slam2_ros2_amd/liborbfe_hip.so
      echo -n "$v: "; timeout -k 10 200 python3 $R/tools/step_time.py $n 100 2>&1 | tail -1
    done
  done
done
cp /tmp/keep.so $R/orb_slam2_ros2_amd/liborbfe_hip.so
