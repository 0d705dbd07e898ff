set -e
for pc in 8 16; do
echo "per_cu $pc"
ORBFE_QT_PER_CU=$pc python3 tools/stage_times.py 512
ORBFE_QT_PER_CU=$pc python3 tools/step_time.py 512 40
done
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_config.py -x -q -m gpu > gpurun_out/r4_t4.txt 2>&1 || { tail -30 gpurun_out/r4_t4.txt; exit 1; }
tail -3 gpurun_out/r4_t4.txt
