#!/bin/bash
# Runs on the GPU box: where does the p99 of the one-frame C++ harness come from?  tests/cpp/test_dropin `latency` (2000 frames per call
# shape) (a) alone on the GPU, (b) beside ONE idle process that holds a HIP context (what bench.py's latency child was until r6: its Python
# parent had already created contexts), (c) beside TWO such processes (bench.py main + the --only-leg child), (d) alone again.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
T=/tmp/lat_ctx; mkdir -p $T
g++ -std=c++17 -O2 -I$R/tests/cpp/stubs -o $T/test_dropin $R/tests/cpp/test_dropin.cpp -L$R/orb_slam2_ros2_amd -lorbfe_hip -pthread -Wl,-rpath,$R/orb_slam2_ros2_amd -Wl,-rpath,/opt/rocm/lib
python3 - <<PY
import sys; sys.path.insert(0, "$R")
from orb_slam2_ros2_amd import synth
L, Rr = synth.stereo_pair(0, 1241, 376)
L.tofile("$T/L.raw"); Rr.tofile("$T/R.raw")
PY
N=${1:-2000}
run() { echo "== $1"; $T/test_dropin latency $T/L.raw $T/R.raw 1241 376 $N | tr ' ' '\n' | paste -sd' ' ; }
idle() { python3 -c "
import sys, time; sys.path.insert(0, '$R')
import torch
from orb_slam2_ros2_amd._lib import Context
torch.zeros(4, device='cuda').sum().item()
c = Context(1241, 376, max_images=2)
open('$T/idle_$1.ready', 'w').write('1')
time.sleep($2)
" & }
run "alone"
rm -f $T/idle_*.ready
idle 1 60; P1=$!
while [ ! -f $T/idle_1.ready ]; do sleep 0.2; done
run "beside one idle HIP process"
idle 2 40; P2=$!
while [ ! -f $T/idle_2.ready ]; do sleep 0.2; done
run "beside two idle HIP processes"
kill $P1 $P2 2>/dev/null || true; wait $P1 $P2 2>/dev/null || true
run "alone again"
# (e) beside a process that looks like bench.py's main process at the time the r5 leg ran: a one-rank RCCL process group (its helper
#     threads), 1 GB of page-locked memory, a closed 1024-slot context
python3 -c "
import os, sys, time; sys.path.insert(0, '$R')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29531'); os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
import torch, torch.distributed as dist
from orb_slam2_ros2_amd._lib import Context
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
t = torch.ones(4, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
pin = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
c = Context(1241, 376, max_images=1024); c.close()
open('$T/idle_3.ready', 'w').write('1')
time.sleep(45)
" > /dev/null 2>&1 &
P3=$!
while [ ! -f $T/idle_3.ready ]; do sleep 0.2; done
N=500
run "beside a bench-like parent (RCCL group of one rank, 1 GB pinned), 500 frames"
run "the same again, 500 frames"
kill $P3 2>/dev/null || true; wait $P3 2>/dev/null || true
N=500
run "alone, 500 frames"
