#!/bin/bash
# runs on the GPU box: step time under a few runtime switches (same library, same box)
for rep in 1 2; do for k in "ORBFE_FAST_SIDE_FROM=3" "ORBFE_FAST_SIDE_FROM=0" "ORBFE_FAST_SIDE_FROM=1" "ORBFE_FAST_SIDE_FROM=2" "ORBFE_FAST_SIDE_FROM=4"; do
  echo -n "[$k]  "; env $k timeout 200 python tools/step_time.py ${1:-512} ${2:-60} 2>/dev/null
done; done
