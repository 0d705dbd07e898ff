#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tl
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/tl -- python3 $R/tools/step_time.py 512 12 > /tmp/tl.log 2>&1
db=$(find /tmp/tl -name '*.db' | head -1)
python3 $R/tools/exp/timeline.py $db
