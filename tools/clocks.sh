#!/bin/bash
# clocks / power of the GPU while the 512^3 case is stepping, next to the step time of that run (which box is this?)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
python3 $ROOT/tools/series.py 512 ${1:-90} smooth > $ROOT/gpurun_out/clocks_series.log 2>&1 &
PID=$!
sleep 9
for s in 1 2 3; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "fclk|mclk|sclk|Power \(W\)|junction|memory\)" | sed 's/^GPU\[0\]\t\t: //' | tr '\n' ';'
  echo
  sleep 0.7
done
wait $PID
tail -3 $ROOT/gpurun_out/clocks_series.log
