#!/bin/bash
# Runs on the GPU box: busy cycles of the vector ALU per kernel against its instruction count (two counter-only passes of the bench step loop)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_THREAD_CYCLES_VALU" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES"; do
  rm -rf $OUT/prof_vc
  rocprofv3 --pmc $set --output-format csv -d $OUT/prof_vc -- python3 bench.py --steps 4 --warmup 1 --prewarm-seconds 0.2 --cpu-seconds 0 --host-io-steps 0 --sequence-leg 0 --legs '' > /dev/null 2> $OUT/prof_vc.err
  Q=$(find $OUT/prof_vc -name "*counter_collection.csv" | head -1)
  python3 tools/pmc_kernels.py $Q
  echo
done
rm -rf $OUT/prof_vc
