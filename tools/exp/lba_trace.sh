#!/bin/bash
# Runs on the GPU box: kernel trace of orbfe_ba_local_optimize on the config-5 problem (tools/lba_time.py) -> gpurun_out/<tag>_lba_kernel_stats.csv
set -e
TAG=${1:-rX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
rm -rf $OUT/prof_lba
rocprofv3 --kernel-trace -d $OUT/prof_lba -- python3 tools/lba_time.py 10 > $OUT/${TAG}_lba_time.txt 2> $OUT/prof_lba.err
DB=$(find $OUT/prof_lba -name "*.db" | head -1)
python3 tools/kernel_stats_from_db.py $DB > $OUT/${TAG}_lba_kernel_stats.csv
python3 - "$DB" > $OUT/${TAG}_lba_timeline.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start, end, name from kernels order by start").fetchall()
# the last complete optimisation: from the last k_lm_maxdiag that is preceded by a gap > 100 us back to the end
fin = [i for i, r in enumerate(rows) if "k_lm_final" in r[2]]
i1 = fin[-1] + 2
i0 = fin[-2] + 2 if len(fin) > 1 else 0
while i0 < i1 and "k_lm_ctrl" in rows[i0][2]: i0 += 1
t0 = rows[i0][0]
prev = t0
for s, e, n in rows[i0:i1]:
    print(f"{(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:6.1f}  {n.split('(')[0].replace('orbfe::', '')}")
    prev = e
PY
cat $OUT/${TAG}_lba_time.txt
cut -c1-150 $OUT/${TAG}_lba_kernel_stats.csv
rm -rf $OUT/prof_lba
