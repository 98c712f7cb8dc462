set -e
( while true; do sleep 50; echo "[alive] $(date +%T)"; done ) &
HB=$!
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "c3_full_size_512" --durations=3 > gpurun_out/r2_t21.log 2>&1 || { kill $HB; tail -40 gpurun_out/r2_t21.log | cut -c1-600; exit 1; }
kill $HB
tail -8 gpurun_out/r2_t21.log
