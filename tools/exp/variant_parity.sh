#!/bin/bash
# runs on the GPU box: parity suite + step time for each variant library
cp orb_slam2_ros2_amd/liborbfe_hip.so /tmp/keep.so
for v in $1; do
  cp tools/exp/libs/liborbfe_$v.so orb_slam2_ros2_amd/liborbfe_hip.so
  echo "== $v: $(timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -1)"
  for r in 1 2; do echo -n "   "; timeout 200 python tools/step_time.py 512 80 2>/dev/null; done
done
cp /tmp/keep.so orb_slam2_ros2_amd/liborbfe_hip.so
