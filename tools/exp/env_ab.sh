#!/bin/bash
# Runs on the GPU box: tools/exp/env_ab.sh VAR "v1 v2 ..." [pairs] [steps] -- step time under each value of one environment switch, alternating, two rounds
R=$GRAFT_REPO_ROOT
for round in 1 2; do
  for v in $2; do
    echo -n "$1=$v: "; env $1=$v timeout -k 10 200 python3 $R/tools/step_time.py ${3:-512} ${4:-120} 2>&1 | tail -1
  done
done
