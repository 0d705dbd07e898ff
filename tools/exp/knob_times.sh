#!/bin/bash
for rep in 1 2 3; do for k in "ORBFE_FAST0_EARLY=0" "ORBFE_FAST0_EARLY=1"; do
  echo -n "[$k]  "; env $k timeout 200 python tools/step_time.py ${1:-512} ${2:-60} 2>/dev/null
done; done
