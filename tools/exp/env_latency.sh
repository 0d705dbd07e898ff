#!/bin/bash
# Runs on the GPU box: tools/latency_single.py and the C++ latency harness under two values of an environment switch, alternating:
#   tools/exp/env_latency.sh ORBFE_HOST_RESIZE_DIRECT "0 1"
cd ${GRAFT_REPO_ROOT:-.}
T=$(mktemp -d)
g++ -std=c++17 -O2 -Itests/cpp/stubs -o $T/test_dropin tests/cpp/test_dropin.cpp -Lorb_slam2_ros2_amd -lorbfe_hip -pthread -Wl,-rpath,$PWD/orb_slam2_ros2_amd -Wl,-rpath,/opt/rocm/lib
python3 - "$T" <<'PY'
import sys; sys.path.insert(0, ".")
from orb_slam2_ros2_amd import synth
L, R = synth.stereo_pair(0); L.tofile(sys.argv[1] + "/L.raw"); R.tofile(sys.argv[1] + "/R.raw")
PY
for round in 1 2 3; do
  for v in $2; do
    echo "== $1=$v"
    env $1=$v timeout -k 10 120 python3 tools/latency_single.py 2>&1 | grep -v amdgpu.ids | head -2
    env $1=$v timeout -k 10 200 $T/test_dropin latency $T/L.raw $T/R.raw 1241 376 1000 2>&1 | grep "LATQ" | cut -c1-120
  done
done
