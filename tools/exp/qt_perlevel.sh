#!/bin/bash
# Runs on the GPU box: the step with the quadtree's per-level node tables / record caches on and off (ORBFE_QT_PER_LEVEL)
# (needs a library built from a patched tree: the experiment was reverted after the measurement recorded in DESIGN 4.9 -- the script documents how it was run)
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for n in 0 1; do
  echo -n "per_level $n: "
  ORBFE_QT_PER_LEVEL=$n python3 bench.py --legs "" --steps 60 --cpu-seconds 0 --host-io-steps 0 --sequence-leg 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], 'qt', d['roofline']['all_stages']['quadtree']['ms'], d['verified_pairs'])"
done
done
