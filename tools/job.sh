set -e
python -m pytest tests/test_hip_parity.py tests/test_golden.py -q -m gpu -x > gpurun_out/r2_t20.log 2>&1 || { tail -40 gpurun_out/r2_t20.log | cut -c1-600; exit 1; }
tail -3 gpurun_out/r2_t20.log
bash tools/trace.sh r02h 12 512 12
