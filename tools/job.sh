set -e
python -m pytest tests/test_hip_parity.py tests/test_golden.py -q -m gpu -x -k "conv_diff or golden or mom_step or noncubic" > gpurun_out/r2_t6.log 2>&1 || { tail -40 gpurun_out/r2_t6.log | cut -c1-400; exit 1; }
tail -3 gpurun_out/r2_t6.log
export WL_PRESTEPS=12
WL_CLASSES=conv_diff,smooth python tools/sweep.py 512 18 1 0 > gpurun_out/r2_sweep_cd2.log 2>&1; cat gpurun_out/r2_sweep_cd2.log
