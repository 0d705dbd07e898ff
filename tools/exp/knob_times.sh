#!/bin/bash
for rep in 1 2 3; do for k in "ORBFE_BLUR_L0_EARLY=1 ORBFE_FAST_SIDE_FROM=0" "ORBFE_BLUR_L0_EARLY=0 ORBFE_FAST_SIDE_FROM=3" "ORBFE_BLUR_L0_EARLY=0 ORBFE_FAST_SIDE_FROM=0"; do
  echo -n "[$k]  "; env $k timeout 200 python tools/step_time.py ${1:-512} ${2:-60} 2>/dev/null
done; done
