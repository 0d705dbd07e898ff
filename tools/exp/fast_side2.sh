#!/bin/bash
# Runs on the GPU box: r6's side stream of FAST's small levels -- default rule / merged into one launch / off / explicit masks, alternating
cd ${GRAFT_REPO_ROOT:-.}
for round in $(seq 1 ${1:-3}); do
  for cls in ${2:-rect camera}; do
    echo -n "off    : "; ORBFE_FAST_SIDE_MASK=0 timeout -k 10 300 python3 tools/ab_content.py $cls 512 150 2>&1 | tail -1 | cut -c1-50
    echo -n "auto   : "; timeout -k 10 300 python3 tools/ab_content.py $cls 512 150 2>&1 | tail -1 | cut -c1-50
    echo -n "merged : "; ORBFE_FAST_SIDE_MERGE=1 timeout -k 10 300 python3 tools/ab_content.py $cls 512 150 2>&1 | tail -1 | cut -c1-50
  done
done
