#!/bin/bash
# Runs on the GPU box: the step time by ORBFE_FAST_SIDE_FROM (levels >= N of k_fast on the second stream) with the library in place
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for n in 0 1 2 3 4; do
  echo -n "side_from $n: "
  ORBFE_FAST_SIDE_FROM=$n python3 bench.py --legs "" --steps 60 --cpu-seconds 0 --host-io-steps 0 --sequence-leg 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], 'fast', d['roofline']['all_stages']['fast']['ms'])"
done
done
