#!/bin/bash
# Runs on the GPU box: one-pair latency with the captured graph and with plain launches (ORBFE_GRAPHS), alternating
R=$GRAFT_REPO_ROOT
for round in 1 2 3; do
  for g in 1 0; do
    echo -n "graphs=$g: "; ORBFE_GRAPHS=$g timeout -k 10 200 python3 $R/tools/latency_single.py 2>&1 | grep "extract_batch ms"
  done
done
