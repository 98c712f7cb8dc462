set -e
python -m pytest tests/test_hip_parity.py tests/test_golden.py tests/test_multi_gpu.py -q -m gpu -x > gpurun_out/r2_t18.log 2>&1 || { tail -40 gpurun_out/r2_t18.log | cut -c1-600; exit 1; }
tail -3 gpurun_out/r2_t18.log
python tools/ab_step.py 512 23 5 > gpurun_out/r2_ab23.log 2>&1; cat gpurun_out/r2_ab23.log
bash tools/trace.sh r02g 8 512 24 > /dev/null
grep -E "k_bc_vec_all|kernel  |k_correct3|bdim" gpurun_out/trace_r02g.txt
