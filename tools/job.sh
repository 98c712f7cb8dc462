set -e
python -m pytest tests/test_hip_parity.py tests/test_golden.py -q -m gpu -x > gpurun_out/r2_t15.log 2>&1 || { tail -40 gpurun_out/r2_t15.log | cut -c1-600; exit 1; }
tail -3 gpurun_out/r2_t15.log
python tools/ab_step.py 512 22 5 > gpurun_out/r2_ab22.log 2>&1; cat gpurun_out/r2_ab22.log
bash tools/trace.sh r02d 16 512 40
