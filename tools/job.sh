set -e
for r in 1 2; do
  for a in 0 1 4 16 64; do
    WL_SPACER_GB=$a python bench.py --steps 10 --warmup 5 --no-cpu-baseline > gpurun_out/sp_${a}_$r.json 2> gpurun_out/sp_${a}_$r.err
    python - <<PY
import json
d=json.load(open("gpurun_out/sp_${a}_$r.json"))
pc=d["roofline"]["per_class_ms_one_step"]
print("spacer=$a GB run $r: %.2f ms/step  smooth %.3f  prolong %.3f conv %.3f correct %.3f bdim %.3f cfl %.3f residual %.3f" % (d["ms_per_step"], d["smoother"]["avg_launch_ms"], d["prolong_increment"]["avg_launch_ms"], pc["conv_diff"]["ms"]/2, pc["correct"]["ms"]/2, pc["bdim"]["ms"]/2, pc["cfl"]["ms"], pc["residual"]["ms"]/2))
PY
  done
done
