#!/bin/bash
# Runs on the GPU box: cells per k_fast wave (ORBFE_FAST_CPW; unset = the launcher's rule) by batch size
cd ${GRAFT_REPO_ROOT:-.}
for b in 16 32 64 128 256; do
  for cpw in auto 1 2 4; do
    echo -n "pairs $b cpw $cpw: "
    if [ $cpw = auto ]; then python3 tools/step_time.py $b 100 2>/dev/null; else ORBFE_FAST_CPW=$cpw python3 tools/step_time.py $b 100 2>/dev/null; fi
  done
done
