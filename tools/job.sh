set -e
python tools/stagger.py 4352 > gpurun_out/r2_stag1.log 2>&1; cat gpurun_out/r2_stag1.log
python tools/stagger.py 33408 > gpurun_out/r2_stag2.log 2>&1; cat gpurun_out/r2_stag2.log
python tools/stagger.py 1048704 > gpurun_out/r2_stag3.log 2>&1; cat gpurun_out/r2_stag3.log
