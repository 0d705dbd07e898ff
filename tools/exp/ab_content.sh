#!/bin/bash
# Runs on the GPU box: tools/exp/ab_content.sh "<variants>" "<classes>" [pairs] [steps] [rounds] -- per variant library (tools/exp/libs/liborbfe_<v>.so)
# and content class: stage times alone + the step, alternating variants; the digests of two variants must agree
R=$GRAFT_REPO_ROOT
cp $R/orb_slam2_ros2_amd/liborbfe_hip.so /tmp/keep.so
for round in $(seq 1 ${5:-2}); do
  for c in $2; do
    for v in $1; do
      cp $R/tools/exp/libs/liborbfe_$v.so $R/orb_slam2_ros2_amd/liborbfe_hip.so
      echo -n "$v: "; timeout -k 10 300 python3 $R/tools/ab_content.py $c ${3:-512} ${4:-60} 2>&1 | tail -1
    done
  done
done
cp /tmp/keep.so $R/orb_slam2_ros2_amd/liborbfe_hip.so
