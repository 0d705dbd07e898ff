#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python3 tools/exp/qt_occ.py 1000 1920 1080 2>/dev/null | tail -1
python3 tools/exp/qt_occ.py 2000 1920 1080 2>/dev/null | tail -1
python3 tools/exp/qt_occ.py 2000 1241 376 2>/dev/null | tail -1
