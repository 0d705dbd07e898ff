#!/bin/bash
# Runs on the GPU box: the C++ drop-in latency mode (tests/cpp/test_dropin.cpp) against GPU_MAX_HW_QUEUES, interleaved repetitions
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
T=$(mktemp -d)
g++ -std=c++17 -O2 -Itests/cpp/stubs -o $T/test_dropin tests/cpp/test_dropin.cpp -Lorb_slam2_ros2_amd -lorbfe_hip -pthread -Wl,-rpath,$R/orb_slam2_ros2_amd -Wl,-rpath,/opt/rocm/lib
python3 -c "
import sys; sys.path.insert(0,'.')
from orb_slam2_ros2_amd import synth
L,R=synth.stereo_pair(0); L.tofile('$T/L.raw'); R.tofile('$T/R.raw')"
for rep in 1 2 3; do
  for q in 4 6 8 16; do
    echo -n "q=$q: "; GPU_MAX_HW_QUEUES=$q $T/test_dropin latency $T/L.raw $T/R.raw 1241 376 500
  done
done
