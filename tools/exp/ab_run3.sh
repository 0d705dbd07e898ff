#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for nf in 1000 2000 2500 3000 4000; do python3 tools/exp/qt_occ.py $nf 2>/dev/null | tail -1; done
