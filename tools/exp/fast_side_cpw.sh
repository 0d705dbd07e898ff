#!/bin/bash
# Runs on the GPU box: cells per wave (ORBFE_FAST_CPW, every launch) with r6's side stream of the small levels
cd ${GRAFT_REPO_ROOT:-.}
for round in $(seq 1 ${1:-2}); do
  for c in 0 2 4 8; do
    echo -n "cpw $c: "; ORBFE_FAST_CPW=$c timeout -k 10 300 python3 tools/ab_content.py rect 512 150 2>&1 | tail -1 | cut -c1-50
  done
done
