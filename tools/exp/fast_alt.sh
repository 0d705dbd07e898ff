#!/bin/bash
# Runs on the GPU box: the step time by ORBFE_FAST_ALT (k_fast's levels spread over two streams: 1 = odd levels on the second)
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for n in 0 1 2 3; do
  echo -n "fast_alt $n: "
  ORBFE_FAST_ALT=$n python3 bench.py --legs "" --steps 60 --cpu-seconds 0 --host-io-steps 0 --sequence-leg 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], 'fast', d['roofline']['all_stages']['fast']['ms'])"
done
done
