#!/bin/bash
# Runs on the GPU box: kernel + copy trace of bench.py's latency leg -> gpurun_out/<tag>_track_chain_timeline.txt (one orbfe_track_local_map call)
set -e
TAG=${1:-rX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
rm -rf $OUT/prof_trk
rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/prof_trk -- python3 bench.py --steps 2 --warmup 1 --prewarm-seconds 0.1 --cpu-seconds 0 --host-io-steps 0 --sequence-leg 0 --legs latency > $OUT/${TAG}_track_bench.json 2> $OUT/prof_trk.err
DB=$(find $OUT/prof_trk -name "*.db" | head -1)
python3 - "$DB" > $OUT/${TAG}_track_chain_timeline.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
rows = [(s, e, n.split('(')[0].replace('orbfe::', '')) for s, e, n in db.execute("select start, end, name from kernels").fetchall()]
mc = [t for t in tabs if t == "memory_copies"]
if mc:
    cols = [r[1] for r in db.execute("pragma table_info(memory_copies)").fetchall()]
    nm = "name" if "name" in cols else cols[0]
    sz = "size" if "size" in cols else None
    for r in db.execute(f"select start, end, {nm}" + (f", {sz}" if sz else "") + " from memory_copies").fetchall():
        rows.append((r[0], r[1], f"copy {r[2]}" + (f" {r[3]} B" if sz else "")))
rows.sort()
import os
anchor = os.environ.get("TRACK_ANCHOR", "k_track_queries")
idx = [i for i, r in enumerate(rows) if anchor in r[2]]
if anchor != "k_track_queries": idx = [i for i in idx if i > [j for j, r in enumerate(rows) if "k_track_queries" in r[2]][-1]]
i0 = idx[len(idx) // 2]
i1 = idx[len(idx) // 2 + 1]
# walk back over the uploads of this call
j = i0
while j > 0 and rows[j - 1][2].startswith("copy") and rows[i0][0] - rows[j - 1][0] < 200000:
    j -= 1
t0 = rows[j][0]
for s, e, n in rows[j:i1]:
    if s - t0 > 900000: break
    print(f"{(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f}  {n}")
PY
cat $OUT/${TAG}_track_chain_timeline.txt
python3 -c "
import json; d=json.loads(open('$OUT/${TAG}_track_bench.json').read().strip().splitlines()[-1]); print(json.dumps(d['latency']['track_local_map'])[:900])"
rm -rf $OUT/prof_trk
