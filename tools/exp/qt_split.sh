#!/bin/bash
# Runs on the GPU box: the step with ORBFE_QT_SPLIT (the chunk's quadtree in two halves, the first half's lists / moments / orientation /
# descriptors on a second stream beside the second half's trees); bench.py checks every pair against the golden digests first
# (needs a library built from a patched tree: the experiment was reverted after the measurement recorded in DESIGN 4.9 -- the script documents how it was run)
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for n in 0 1; do
  echo -n "qt_split $n: "
  ORBFE_QT_SPLIT=$n python3 bench.py --legs "" --steps 60 --cpu-seconds 0 --host-io-steps 0 --sequence-leg 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['verified_pairs'])"
done
done
