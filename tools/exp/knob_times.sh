#!/bin/bash
for rep in 1 2; do for k in "ORBFE_STEREO_PRIO=0" "ORBFE_STEREO_PRIO=1" "ORBFE_STEREO_PRIO=-1"; do
  echo -n "[$k]  "; env $k timeout 200 python tools/step_time.py ${1:-512} ${2:-60} 2>/dev/null
done; done
