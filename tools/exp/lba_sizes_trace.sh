#!/bin/bash
# Runs on the GPU box: kernel trace of orbfe_ba_local_optimize at the window sizes given (tools/lba_sizes.py) -> per-kernel totals
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
cd $R
rm -rf /tmp/prof_lba
rocprofv3 --kernel-trace -d /tmp/prof_lba -- python3 tools/lba_sizes.py "$@" --no-oracle > /tmp/lba_sizes.txt 2> /tmp/prof_lba.err || { tail -5 /tmp/prof_lba.err; exit 1; }
grep -v amdgpu.ids /tmp/lba_sizes.txt
DB=$(find /tmp/prof_lba -name "*.db" | head -1)
python3 tools/kernel_stats_from_db.py $DB | cut -c1-140 | sed 's/orbfe:://' | head -24
python3 - "$DB" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start, end, name from kernels order by start").fetchall()
# the last trial that really ran: the last k_lmb_back longer than 20 us and the k_lmb_step launches before it
backs = [i for i, r in enumerate(rows) if "k_lmb_back" in r[2] and r[1] - r[0] > 20000]
if backs:
    i1 = backs[-1]
    i0 = i1
    while i0 > 0 and "k_lmb_step" in rows[i0 - 1][2]: i0 -= 1
    d = [(rows[i][1] - rows[i][0]) / 1e3 for i in range(i0, i1)]
    g = [(rows[i + 1][0] - rows[i][1]) / 1e3 for i in range(i0, i1)]
    print("k_lmb_step durations (us) of one trial:", " ".join(f"{x:.1f}" for x in d))
    print("gaps to the next launch (us):", " ".join(f"{x:.1f}" for x in g))
    print("k_lmb_back (us):", (rows[i1][1] - rows[i1][0]) / 1e3)
PY
