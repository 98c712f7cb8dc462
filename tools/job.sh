set -e
( while true; do sleep 50; echo "[alive] $(date +%T)"; done ) &
HB=$!
python -m pytest tests -q -m gpu -x > gpurun_out/r2_full_f.log 2>&1 || { kill $HB; tail -40 gpurun_out/r2_full_f.log | cut -c1-600; exit 1; }
kill $HB
tail -3 gpurun_out/r2_full_f.log
for c in "--size 512" "--size 256"; do
  python bench.py $c --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/cfg.json 2> gpurun_out/cfg.err || { tail -5 gpurun_out/cfg.err; continue; }
  python - "$c" <<'PY'
import json,sys
d=json.load(open("gpurun_out/cfg.json"))
print("%-28s %.2f ms/step  %.0f MLUPS  smoother %.3f ms frac %.3f  n=%s" % (sys.argv[1], d["ms_per_step"], d["value"], d["smoother"]["avg_launch_ms"], d["smoother"]["frac"], d["config"]["vcycles_per_solve"][-2:]))
PY
done
