set -e
python -m pytest tests/test_hip_parity.py tests/test_golden.py -q -m gpu -x > gpurun_out/r2_t8.log 2>&1 || { tail -40 gpurun_out/r2_t8.log | cut -c1-400; exit 1; }
tail -3 gpurun_out/r2_t8.log
export WL_PRESTEPS=12
WL_CLASSES=div,cfl,scale,restrict,dot,correct python tools/sweep.py 512 5 2 > gpurun_out/r2_sweep_dc.log 2>&1; cat gpurun_out/r2_sweep_dc.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench9.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r2_bench9.json')); print(d['ms_per_step'], d['value']); print(d['roofline']['per_class_ms_one_step'])"
