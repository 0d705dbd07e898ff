#!/bin/bash
# Runs on the GPU box: the parts of the reference-shaped frame (two extract() threads + searchByStereo), p50 / p99 and the slowest frames
R=${GRAFT_REPO_ROOT:-.}
cd $R
python3 -c "
import sys; sys.path.insert(0,'.')
from orb_slam2_ros2_amd import synth
L,Rr=synth.stereo_pair(0); L.tofile('/tmp/L.raw'); Rr.tofile('/tmp/R.raw')"
g++ -std=c++17 -O2 -Itests/cpp/stubs -o /tmp/test_dropin tests/cpp/test_dropin.cpp -Lorb_slam2_ros2_amd -lorbfe_hip -pthread -Wl,-rpath,$R/orb_slam2_ros2_amd -Wl,-rpath,/opt/rocm/lib
/tmp/test_dropin latency_tail /tmp/L.raw /tmp/R.raw 1241 376 ${1:-2000}
/tmp/test_dropin latency /tmp/L.raw /tmp/R.raw 1241 376 500
