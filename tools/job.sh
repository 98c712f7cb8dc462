set -e
(rocm-smi --showclocks --showperflevel --showpower --showmaxpower --showmemorypartition --showcomputepartition --showtemp 2>&1 | head -60) > gpurun_out/smi_before.log || true
python tools/series.py 512 24 pcg_direction > gpurun_out/series_c.log 2>&1; tail -4 gpurun_out/series_c.log
(rocm-smi --showclocks --showpower --showtemp 2>&1 | head -40) > gpurun_out/smi_after.log || true
cat gpurun_out/smi_before.log | cut -c1-160
echo ---- ; cat gpurun_out/smi_after.log | cut -c1-160
