set -e
for pc in 8 16; do for rc in -1 0; do
echo "per_cu $pc rec_cap $rc"
if [ $rc = 0 ]; then export ORBFE_QT_REC_CAP=0; else unset ORBFE_QT_REC_CAP; fi
ORBFE_QT_PER_CU=$pc python3 tools/stage_times.py 512
ORBFE_QT_PER_CU=$pc python3 tools/step_time.py 512 40
done; done
