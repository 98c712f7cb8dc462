set -e
python -m pytest tests/test_hip_parity.py tests/test_golden.py -q -m gpu -x > gpurun_out/r2_t13.log 2>&1 || { tail -40 gpurun_out/r2_t13.log | cut -c1-600; exit 1; }
tail -3 gpurun_out/r2_t13.log
export WL_PRESTEPS=12
WL_CLASSES=conv_diff,bdim python tools/sweep.py 512 20 1 0 > gpurun_out/r2_sweep_by8.log 2>&1; cat gpurun_out/r2_sweep_by8.log
python tools/ab_step.py 512 20 4 > gpurun_out/r2_ab20.log 2>&1; cat gpurun_out/r2_ab20.log
