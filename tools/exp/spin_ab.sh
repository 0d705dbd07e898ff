#!/bin/bash
# Runs on the GPU box: the C++ latency harness (tests/cpp/test_dropin latency, 1000 frames per shape) with and without the polling wait
# (ORBFE_SPIN_WAIT=0: hipStreamSynchronize), alternating
cd ${GRAFT_REPO_ROOT:-.}
T=$(mktemp -d)
g++ -std=c++17 -O2 -Itests/cpp/stubs -o $T/test_dropin tests/cpp/test_dropin.cpp -Lorb_slam2_ros2_amd -lorbfe_hip -pthread -Wl,-rpath,$PWD/orb_slam2_ros2_amd -Wl,-rpath,/opt/rocm/lib
python3 - "$T" <<'PY'
import sys; sys.path.insert(0, ".")
from orb_slam2_ros2_amd import synth
L, R = synth.stereo_pair(0); L.tofile(sys.argv[1] + "/L.raw"); R.tofile(sys.argv[1] + "/R.raw")
PY
for round in 1 2 3; do
  for sw in 0 1; do
    echo "== ORBFE_SPIN_WAIT=$sw"; ORBFE_SPIN_WAIT=$sw timeout -k 10 200 $T/test_dropin latency $T/L.raw $T/R.raw 1241 376 1000 2>&1 | grep "LATQ" | cut -c1-150
  done
done
