#!/bin/bash
# Runs on the GPU box: kernel trace of the C++ drop-in latency mode with 4 and 16 hardware queues -> gpurun_out/latq_<q>.txt
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
export TMPDIR=/tmp
T=$(mktemp -d)
g++ -std=c++17 -O2 -Itests/cpp/stubs -o $T/test_dropin tests/cpp/test_dropin.cpp -Lorb_slam2_ros2_amd -lorbfe_hip -pthread -Wl,-rpath,$R/orb_slam2_ros2_amd -Wl,-rpath,/opt/rocm/lib
python3 -c "
import sys; sys.path.insert(0,'.')
from orb_slam2_ros2_amd import synth
L,R=synth.stereo_pair(0); L.tofile('$T/L.raw'); R.tofile('$T/R.raw')"
for q in 4 16; do
  rm -rf gpurun_out/prof_latq
  GPU_MAX_HW_QUEUES=$q rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/prof_latq -- $T/test_dropin latency $T/L.raw $T/R.raw 1241 376 100 > gpurun_out/latq_$q.out 2> gpurun_out/latq_$q.err
  DB=$(find gpurun_out/prof_latq -name "*.db" | head -1)
  python3 - "$DB" > gpurun_out/latq_$q.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("#", cols)
qc = "queue_id" if "queue_id" in cols else "0"
sc = "stream_id" if "stream_id" in cols else "0"
rows = db.execute(f"select start, end, name, {qc}, {sc} from kernels order by start").fetchall()
cp = db.execute("select start, end, name, size, queue_id, stream_id from memory_copies order by start").fetchall()
ev = [(s, e, n.split('(')[0].replace('orbfe::', ''), q, st) for s, e, n, q, st in rows] + [(s, e, f"COPY {n} {sz}", q, st) for s, e, n, sz, q, st in cp]
ev.sort()
idx = [i for i, r in enumerate(ev) if "k_stereo" in r[2]]
for which in (len(idx) // 4, (3 * len(idx)) // 4 + 10):   # a frame of the two-thread loop, a frame of the one-thread loop
    i1 = idx[min(which, len(idx) - 1)]
    i0 = idx[min(which, len(idx) - 1) - 1] + 1
    t0 = ev[i0][0]
    print(f"--- frame ending at k_stereo #{which}")
    for s, e, n, q, st in ev[i0:i1 + 1]:
        print(f"{(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f}  q{q} s{st}  {n}")
PY
  cat gpurun_out/latq_$q.out
done
rm -rf gpurun_out/prof_latq
