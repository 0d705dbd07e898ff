#!/bin/bash
# Same-box A/B of compile-time variants: tools/exp/ab_build.sh <name> "<extra hipcc flags>" builds tools/exp/libs/liborbfe_<name>.so from
# the current sources (all objects rebuilt with the flags).  Run the variants inside ONE gpurun call (boxes differ by ~1.5 %):
#   for v in a b; do cp tools/exp/libs/liborbfe_$v.so orb_slam2_ros2_amd/liborbfe_hip.so; python tools/step_time.py 512 60; done
set -e
cd "$(dirname "$0")/../../orb_slam2_ros2_amd/csrc"
mkdir -p ../../tools/exp/libs /tmp/ab_$1
FLAGS="-std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -I../../include $2"
for f in $(grep '^SRCS' Makefile | sed 's/SRCS = //; s/\.hip//g'); do
  X=""; [ $f = k_fast ] && X="-mllvm -amdgpu-atomic-optimizer-strategy=None"; [ $f = k_blur_mfma ] && X="-mllvm -amdgpu-mfma-vgpr-form"
  /opt/rocm/bin/hipcc $FLAGS $X -c $f.hip -o /tmp/ab_$1/$f.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp/libs/liborbfe_$1.so /tmp/ab_$1/*.o
echo built tools/exp/libs/liborbfe_$1.so
