set -e
python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "update_of_changed or native_measure or moving or x_ghost or fused" > gpurun_out/r2_t19.log 2>&1 || { tail -40 gpurun_out/r2_t19.log | cut -c1-600; exit 1; }
tail -3 gpurun_out/r2_t19.log
python tools/remeasure.py > gpurun_out/r2_remeasure_e.log 2>&1 || true
tail -6 gpurun_out/r2_remeasure_e.log
for c in "--size 256" "--dtype f64" "--grid 1024 1024 512" "--dtype f64 --body donut"; do
  python bench.py $c --steps 10 --warmup 5 --no-cpu-baseline > gpurun_out/cfg.json 2> gpurun_out/cfg.err || { tail -5 gpurun_out/cfg.err; continue; }
  python - "$c" <<'PY'
import json,sys
d=json.load(open("gpurun_out/cfg.json"))
print("%-28s %.2f ms/step  %.0f MLUPS  smoother %.3f ms frac %.3f  n=%s" % (sys.argv[1], d["ms_per_step"], d["value"], d["smoother"]["avg_launch_ms"], d["smoother"]["frac"], d["config"]["vcycles_per_solve"][-2:]))
PY
done
