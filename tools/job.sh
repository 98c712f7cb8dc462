set -e
python -m pytest tests/test_hip_parity.py tests/test_golden.py tests/test_fullsize_properties.py -q -m gpu -x > gpurun_out/r2_t10.log 2>&1 || { tail -40 gpurun_out/r2_t10.log | cut -c1-600; exit 1; }
tail -3 gpurun_out/r2_t10.log
python tools/ab_step.py 512 19 4 > gpurun_out/r2_ab19.log 2>&1; cat gpurun_out/r2_ab19.log
export WL_PRESTEPS=12
WL_CLASSES=bc,conv_diff python tools/sweep.py 512 19 1 0 > gpurun_out/r2_sweep_19.log 2>&1; cat gpurun_out/r2_sweep_19.log
