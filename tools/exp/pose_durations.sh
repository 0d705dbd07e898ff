#!/bin/bash
# Runs on the GPU box: durations of every k_pose_only_reg launch of bench.py's latency leg, in launch order (kernel trace)
R=${GRAFT_REPO_ROOT:-.}
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd $R
rm -rf $OUT/prof_pd
rocprofv3 --kernel-trace -d $OUT/prof_pd -- python3 bench.py --steps 2 --warmup 1 --prewarm-seconds 0.1 --cpu-seconds 0 --host-io-steps 0 --sequence-leg 0 --legs latency > /dev/null 2> $OUT/prof_pd.err
DB=$(find $OUT/prof_pd -name "*.db" | head -1)
python3 - "$DB" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start, end, name from kernels order by start").fetchall()
prev = None
run = []
for i, (s, e, n) in enumerate(rows):
    if "k_pose_only_reg<512>" in n or ("k_pose_only_reg<256>" in n and (e - s) > 20000):
        before = rows[i - 2][2].split("(")[0].replace("orbfe::", "") if i >= 2 else ""
        run.append(((e - s) / 1e3, n.split("(")[0][-24:], before))
import itertools
for k, g in itertools.groupby(run, key=lambda t: (t[1], t[2])):
    g = list(g)
    d = sorted(x[0] for x in g)
    print(f"{k[0]:26s} after {k[1]:28s} x{len(g):4d}  median {d[len(d)//2]:7.1f} us  min {d[0]:7.1f}  max {d[-1]:7.1f}")
PY
rm -rf $OUT/prof_pd
