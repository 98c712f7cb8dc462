set -e
python -m pytest tests -q -m gpu -x > gpurun_out/r2_t7.log 2>&1 || { tail -40 gpurun_out/r2_t7.log | cut -c1-400; exit 1; }
tail -3 gpurun_out/r2_t7.log
for cfg in "--size 512" "--size 256" "--size 512 --dtype f64" "--grid 1024 1024 512" "--size 512 --dtype f64 --body donut"; do
  tag=$(echo $cfg | tr -d ' -')
  python bench.py $cfg --steps 10 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench_$tag.json 2> gpurun_out/r2_bench_$tag.err || { tail -5 gpurun_out/r2_bench_$tag.err; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r2_bench_$tag.json"))
r=d['roofline']; s=d.get('smoother') or {}
print("$cfg", '| ms/step', round(d['ms_per_step'],2), 'MLUPS', round(d['value']), 'vcyc', d['config']['vcycles_per_solve'], '| dominant', r['kernel'], round(r['avg_launch_ms'],3), 'frac', round(r['frac'],3), '| smoother ms', round(s.get('avg_launch_ms',0),3), 'frac', round(s.get('frac',0),3))
PY
done
