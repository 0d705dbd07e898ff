#!/bin/bash
# Runs on the GPU box: bench.py's own step loop (512 DISTINCT frames, 300 steps, no legs) under two environments, alternating:
#   tools/exp/bench_ab_env.sh "ORBFE_FAST_SIDE_MASK=0" [rounds]      (the second arm is the default environment)
cd ${GRAFT_REPO_ROOT:-.}
for round in $(seq 1 ${2:-3}); do
  for arm in "$1" "_DEFAULT=1"; do
    echo -n "$arm : "
    env $arm timeout -k 10 400 python3 bench.py --steps 300 --cpu-seconds 0 --host-io-steps 0 --sequence-leg 0 --legs '' --content-steps 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value']), round(d['ms_per_step'],3), d['verified_pairs'], 'fast live', d['config'].get('stage_fast_live_ms'))"
  done
done
