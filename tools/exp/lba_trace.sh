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
cat $OUT/${TAG}_lba_time.txt
cut -c1-150 $OUT/${TAG}_lba_kernel_stats.csv
rm -rf $OUT/prof_lba
