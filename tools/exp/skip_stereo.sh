#!/bin/bash
# Runs on the GPU box: what would a FREE stereo match buy?  The step with the match kernel skipped (results invalid: timing only) -- the
# bound on anything done to k_stereo.  Needs tools/exp/libs/liborbfe_skipst.so (a build whose run_stereo honours ORBFE_EXP_SKIP_STEREO).
cd ${GRAFT_REPO_ROOT:-.}
cp orb_slam2_ros2_amd/liborbfe_hip.so /tmp/keep.so
cp tools/exp/libs/liborbfe_skipst.so orb_slam2_ros2_amd/liborbfe_hip.so
for rep in 1 2; do
  echo -n "with the match: "; python3 tools/step_time.py 512 60
  echo -n "match skipped:  "; ORBFE_EXP_SKIP_STEREO=1 python3 tools/step_time.py 512 60
done
cp /tmp/keep.so orb_slam2_ros2_amd/liborbfe_hip.so
