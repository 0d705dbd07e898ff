set -e
mkdir -p gpurun_out
python3 tools/step_time.py 512 40 > gpurun_out/r4_base_step.txt 2>&1
python3 tools/stage_times.py 512 >> gpurun_out/r4_base_step.txt 2>&1
bash tools/exp/qt_stamps.sh > gpurun_out/r4_qt_stamps_single.txt 2>&1
QT_BATCH=512 bash tools/exp/qt_stamps.sh > gpurun_out/r4_qt_stamps_batch.txt 2>&1
cat gpurun_out/r4_base_step.txt
