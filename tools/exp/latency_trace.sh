#!/bin/bash
# Runs on the GPU box: kernel trace of the single-pair path (tools/latency_single.py) -> gpurun_out/<tag>_latency_kernel_stats.csv + timeline of one pair
set -e
TAG=${1:-rX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
python3 tools/latency_single.py > $OUT/${TAG}_latency_single.txt 2>&1
rm -rf $OUT/prof_lat
rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/prof_lat -- python3 tools/latency_single.py > /dev/null 2> $OUT/prof_lat.err
DB=$(find $OUT/prof_lat -name "*.db" | head -1)
python3 tools/kernel_stats_from_db.py $DB > $OUT/${TAG}_latency_kernel_stats.csv
python3 - "$DB" > $OUT/${TAG}_latency_timeline.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start, end, name from kernels order by start").fetchall()
# one pair in the middle of the first (batched) loop: from a k_resize_regions/k_load_level0 start to the k_stereo end
idx = [i for i, r in enumerate(rows) if "k_stereo" in r[2]]
i1 = idx[len(idx) // 4]
i0 = i1
while i0 > 0 and "k_stereo" not in rows[i0 - 1][2]:
    i0 -= 1
t0 = rows[i0][0]
for s, e, n in rows[i0:i1 + 1]:
    print(f"{(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f}  {n.split('(')[0].replace('orbfe::', '')}")
PY
cat $OUT/${TAG}_latency_single.txt
cat $OUT/${TAG}_latency_timeline.txt
rm -rf $OUT/prof_lat
