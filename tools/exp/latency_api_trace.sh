#!/bin/bash
# Runs on the GPU box: HIP API + kernel + copy trace of the single-pair path -> gpurun_out/<tag>_latency_api_timeline.txt (one pair, host and device rows interleaved)
set -e
TAG=${1:-rX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
cat > /tmp/lat_pairs.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
from orb_slam2_ros2_amd import synth
from orb_slam2_ros2_amd._lib import Context
L, R = synth.stereo_pair(0)
ctx = Context(1241, 376, max_images=2)
for _ in range(60):
    ctx.extract_batch([L, R]); ctx.stereo_match(0, 1, 718.856, 386.14)
PY
rm -rf $OUT/prof_api
rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace -d $OUT/prof_api -- python3 /tmp/lat_pairs.py > /dev/null 2> $OUT/prof_api.err
DB=$(find $OUT/prof_api -name "*.db" | head -1)
python3 - "$DB" > $OUT/${TAG}_latency_api_timeline.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
rows = [(s, e, "GPU  " + n.split('(')[0].replace('orbfe::', '')) for s, e, n in db.execute("select start, end, name from kernels").fetchall()]
if "memory_copies" in tabs:
    cols = [r[1] for r in db.execute("pragma table_info(memory_copies)").fetchall()]
    nm = "name" if "name" in cols else cols[0]
    for r in db.execute(f"select start, end, {nm}" + (", size" if "size" in cols else ", 0") + " from memory_copies").fetchall():
        rows.append((r[0], r[1], f"DMA  {r[2]} {r[3]} B"))
api = None
for t in ("regions", "regions_and_samples"):
    if t in tabs:
        api = t
        break
if api:
    cols = [r[1] for r in db.execute(f"pragma table_info({api})").fetchall()]
    for r in db.execute(f"select start, end, name from {api}").fetchall():
        rows.append((r[0], r[1], "host " + str(r[2])))
else:
    print("tables:", tabs)
rows.sort()
idx = [i for i, r in enumerate(rows) if "k_stereo" in r[2]]
i1 = idx[len(idx) // 2]
i0 = idx[len(idx) // 2 - 1] + 1
t0 = rows[i0][0]
for s, e, n in rows[i0:i1 + 12]:
    print(f"{(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f}  {n}")
PY
cat $OUT/${TAG}_latency_api_timeline.txt
rm -rf $OUT/prof_api
