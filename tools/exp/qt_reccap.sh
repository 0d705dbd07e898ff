#!/bin/bash
# Runs on the GPU box: the step time by ORBFE_QT_REC_CAP (candidate records of a tree cached in LDS; fewer trees per CU) and ORBFE_QT_GROUPS
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for kv in "X=0" "ORBFE_QT_REC_CAP=600" "ORBFE_QT_REC_CAP=1500" "ORBFE_QT_REC_CAP=3000" "ORBFE_QT_GROUPS=4" "ORBFE_QT_GROUPS=8"; do
  echo -n "$kv: "
  env $kv python3 bench.py --legs "" --steps 60 --cpu-seconds 0 --host-io-steps 0 --sequence-leg 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], 'qt', d['roofline']['all_stages']['quadtree']['ms'])"
done
done
