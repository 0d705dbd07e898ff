#!/bin/bash
# Runs on the GPU box (gpurun): tools/profile_round.sh <tag>  ->  gpurun_out/<tag>_bench.json, <tag>_kernel_stats.csv, pmc_traffic.json
# (copy the three into profiles/ afterwards).  Kernel trace and the two PMC passes are separate rocprofv3 runs of the same command.
set -e
TAG=${1:-rX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout -k 10 600 python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_bench.json; echo
rm -rf $OUT/prof_kt $OUT/prof_f $OUT/prof_w
timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT/prof_kt -- python3 bench.py --steps 40 --cpu-seconds 0 --host-io-steps 0 --sequence-leg 0 --legs '' > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/prof_kt.err
DB=$(find $OUT/prof_kt -name "*.db" | head -1)
python3 tools/kernel_stats_from_db.py $DB > $OUT/${TAG}_kernel_stats.csv
head -4 $OUT/${TAG}_kernel_stats.csv | cut -c1-160
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_f -- python3 bench.py --steps 8 --warmup 1 --prewarm-seconds 0.2 --cpu-seconds 0 --host-io-steps 0 --sequence-leg 0 --legs '' > /dev/null 2> $OUT/prof_f.err
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_w -- python3 bench.py --steps 8 --warmup 1 --prewarm-seconds 0.2 --cpu-seconds 0 --host-io-steps 0 --sequence-leg 0 --legs '' > /dev/null 2> $OUT/prof_w.err
F=$(find $OUT/prof_f -name "*counter_collection.csv" | head -1)
W=$(find $OUT/prof_w -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $F $W 512 $OUT/pmc_traffic.json 1024 profiles/r3_fetch_calibration.json
python3 -c "import json; d=json.load(open('$OUT/pmc_traffic.json')); print({k: round(v['hbm_bytes_per_launch']/1e9, 3) for k, v in d['kernels'].items()})"
# per-wave instruction counts (their own pass: counters only, no trace domains)
rm -rf $OUT/prof_sq
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/prof_sq -- python3 bench.py --steps 6 --warmup 1 --prewarm-seconds 0.2 --cpu-seconds 0 --host-io-steps 0 --sequence-leg 0 --legs '' > /dev/null 2> $OUT/prof_sq.err
Q=$(find $OUT/prof_sq -name "*counter_collection.csv" | head -1)
python3 tools/sq_summary.py $Q $OUT/${TAG}_sq_counters.json "rocprofv3 --pmc SQ_* pass of python3 bench.py (512 pairs per step), tools/profile_round.sh $TAG" 512 > /dev/null || echo "sq counters failed"
rm -rf $OUT/prof_sq
# keep the merge small: the raw traces stay on the box
rm -rf $OUT/prof_kt $OUT/prof_f $OUT/prof_w
