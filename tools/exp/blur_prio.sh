#!/bin/bash
# Runs on the GPU box: the step by the blur stream's priority (ORBFE_BLUR_PRIO: 1 = highest, -1 = lowest, 0 = default)
# (needs a library built from a patched tree: the experiment was reverted after the measurement recorded in DESIGN 4.9 -- the script documents how it was run)
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for n in 0 1 -1; do
  echo -n "blur_prio $n: "
  ORBFE_BLUR_PRIO=$n python3 bench.py --legs "" --steps 60 --cpu-seconds 0 --host-io-steps 0 --sequence-leg 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'])"
done
done
