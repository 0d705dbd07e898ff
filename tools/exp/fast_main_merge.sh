#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for round in $(seq 1 ${1:-3}); do
  echo -n "separate: "; timeout -k 10 300 python3 tools/ab_content.py rect 512 150 2>&1 | tail -1 | cut -c1-120
  echo -n "merged  : "; ORBFE_FAST_MAIN_MERGE=1 timeout -k 10 300 python3 tools/ab_content.py rect 512 150 2>&1 | tail -1 | cut -c1-120
  echo -n "m 0xfc  : "; ORBFE_FAST_SIDE_MASK=0xfc ORBFE_FAST_MAIN_MERGE=1 timeout -k 10 300 python3 tools/ab_content.py rect 512 150 2>&1 | tail -1 | cut -c1-120
done
