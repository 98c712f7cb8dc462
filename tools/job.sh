set -e
python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "native or geometry or sphere_96 or c1_full" > gpurun_out/r2_t4.log 2>&1 || { tail -40 gpurun_out/r2_t4.log | cut -c1-400; exit 1; }
tail -3 gpurun_out/r2_t4.log
python tools/remeasure.py > gpurun_out/r2_remeasure.log 2>&1 || tail -20 gpurun_out/r2_remeasure.log
cat gpurun_out/r2_remeasure.log
