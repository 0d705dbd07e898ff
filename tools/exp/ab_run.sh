#!/bin/bash
# runs on the GPU box: tools/exp/ab_run.sh "<variants>" [pairs] [steps] [reps]
cp orb_slam2_ros2_amd/liborbfe_hip.so /tmp/keep.so
for rep in $(seq 1 ${4:-2}); do for v in $1; do cp tools/exp/libs/liborbfe_$v.so orb_slam2_ros2_amd/liborbfe_hip.so; echo -n "$v  "; timeout 200 python tools/step_time.py ${2:-512} ${3:-60} 2>/dev/null; done; done
cp /tmp/keep.so orb_slam2_ros2_amd/liborbfe_hip.so
