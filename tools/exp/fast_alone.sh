#!/bin/bash
# kernel trace of the bench loop with the schedule in which nothing runs beside FAST (level-0 blur after FAST, small levels on the second
# stream): the same kernels as the production schedule, k_fast's union time per step without company
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp ORBFE_BLUR_L0_EARLY=0 ORBFE_FAST_SIDE_FROM=3
cd $R
rm -rf /tmp/fa
rocprofv3 --kernel-trace -d /tmp/fa -- python3 bench.py --steps 40 --cpu-seconds 0 --host-io-steps 0 --sequence-leg 0 > $R/gpurun_out/fast_alone_bench.json 2> /tmp/fa.err
DB=$(find /tmp/fa -name "*.db" | head -1)
python3 tools/kernel_stats_from_db.py $DB > $R/gpurun_out/r2_v11_kernel_stats_fast_unaccompanied.csv
head -3 $R/gpurun_out/r2_v11_kernel_stats_fast_unaccompanied.csv | cut -c1-60,170-400
tail -c 300 $R/gpurun_out/fast_alone_bench.json
