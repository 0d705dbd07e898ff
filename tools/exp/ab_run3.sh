#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for nf in 500 800 1000 1200 1600 2000 3000; do python3 tools/exp/qt_occ.py $nf 2>/dev/null | tail -1; done
