set -e
WL_CLASSES=pcg_mult_dot,pcg_update,pcg_direction,smooth,residual,correct,bdim python tools/sweep.py 512 4 1 2 > gpurun_out/r2_sweep2.log 2>&1
cat gpurun_out/r2_sweep2.log
python -m pytest tests/test_hip_parity.py tests/test_golden.py -q -m gpu > gpurun_out/r2_t2.log 2>&1 || { tail -40 gpurun_out/r2_t2.log | cut -c1-300; }
tail -3 gpurun_out/r2_t2.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2_bench2.json 2> gpurun_out/r2_bench2.err
cut -c1-600 gpurun_out/r2_bench2.json
