set -e
python -m pytest tests -q -m gpu -x > gpurun_out/r2_full_d.log 2>&1 || { tail -40 gpurun_out/r2_full_d.log | cut -c1-600; exit 1; }
tail -3 gpurun_out/r2_full_d.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/profile.sh r02d > gpurun_out/prof_r02d.log 2>&1; tail -48 gpurun_out/prof_r02d.log
python bench.py --steps 20 --warmup 5 > gpurun_out/prof_r02d/bench_plain.json 2> gpurun_out/prof_r02d/bench_plain.err
