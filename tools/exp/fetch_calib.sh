#!/bin/bash
# Runs on the GPU box: the FETCH_SIZE / WRITE_SIZE calibration -> gpurun_out/r3_fetch_calibration.json (copy into profiles/)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
rm -rf $OUT/cal_f $OUT/cal_w
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_f -- tools/exp/bin/fetch_calib > $OUT/cal_true.txt 2> $OUT/cal_f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_w -- tools/exp/bin/fetch_calib > /dev/null 2> $OUT/cal_w.err
F=$(find $OUT/cal_f -name "*counter_collection.csv" | head -1)
W=$(find $OUT/cal_w -name "*counter_collection.csv" | head -1)
python3 tools/exp/fetch_calib_summary.py $F $W $OUT/cal_true.txt $OUT/r3_fetch_calibration.json
rm -rf $OUT/cal_f $OUT/cal_w
