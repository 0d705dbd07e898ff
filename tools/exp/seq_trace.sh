#!/bin/bash
# Runs on the GPU box: kernel + memory-copy trace of the 4541-pair sequence job -> gpurun_out/<tag>_seq_timeline.txt (copies and per-batch kernel spans)
set -e
TAG=${1:-rX}
shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
rm -rf $OUT/prof_seq
rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/prof_seq -- python3 bench.py --sequence 4541 "$@" > $OUT/${TAG}_seq_traced.json 2> $OUT/prof_seq.err
DB=$(find $OUT/prof_seq -name "*.db" | head -1)
python3 - "$DB" > $OUT/${TAG}_seq_timeline.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
mc = [n for n in names if "memory_cop" in n and not n.startswith("rocpd_")]
print("# views:", [n for n in names if not n.startswith("rocpd_")][:40])
cols = [r[1] for r in db.execute(f"pragma table_info({mc[0]})")]
print("# copy columns:", cols)
ev = []
size_col = "size" if "size" in cols else [c for c in cols if "size" in c or "bytes" in c][0]
for s, e, n, sz in db.execute(f"select start, end, name, {size_col} from {mc[0]} where {size_col} > 1000000 order by start"):
    ev.append((s, e, f"COPY {n} {sz / 1e6:.1f} MB  {sz / max(1, e - s):.1f} GB/s"))
rows = db.execute("select start, end, name from kernels order by start").fetchall()
# kernel spans: group kernels separated by < 200 us
grp = None
for s, e, n in rows:
    if grp and s - grp[1] < 200e3:
        grp[1] = max(grp[1], e); grp[2] += 1; grp[3] += e - s
    else:
        if grp: ev.append((grp[0], grp[1], f"KERNELS x{grp[2]} busy {grp[3] / 1e6:.2f} ms"))
        grp = [s, e, 1, e - s]
if grp: ev.append((grp[0], grp[1], f"KERNELS x{grp[2]} busy {grp[3] / 1e6:.2f} ms"))
ev.sort()
big = [x for x in ev if x[2].startswith("COPY") and "478" in x[2] or "MB" in x[2]]
# the timed job = the last 140 ms before the last big copy ends
tend = max(e for s, e, _ in ev)
t0 = min(s for s, e, _ in ev if s > tend - 400e6)
for s, e, d in ev:
    if s >= t0:
        print(f"{(s - t0) / 1e6:9.3f} .. {(e - t0) / 1e6:9.3f} ms  {(e - s) / 1e6:8.3f}  {d}")
PY
tail -80 $OUT/${TAG}_seq_timeline.txt
rm -rf $OUT/prof_seq
