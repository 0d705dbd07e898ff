#!/bin/bash
# Runs on the GPU box: the k_fast launches of some levels on a stream of their own (ORBFE_FAST_SIDE_MASK, bit l = level l), alternating with
# the default, same library: tools/exp/fast_side.sh "0 0xf0 0xaa 0xe0 0x0e" [rounds]
cd ${GRAFT_REPO_ROOT:-.}
for round in $(seq 1 ${2:-3}); do
  for m in $1; do
    echo -n "mask $m: "; ORBFE_FAST_SIDE_MASK=$m timeout -k 10 300 python3 tools/ab_content.py ${3:-rect} 512 150 2>&1 | tail -1
  done
done
