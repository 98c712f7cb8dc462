set -e
export WL_PRESTEPS=12
WL_CLASSES=pcg_mult_dot,pcg_update,pcg_direction,smooth,prolongate,residual,bdim python tools/sweep.py 512 16 4 8 16 > gpurun_out/r2_sweep_g16.log 2>&1; cat gpurun_out/r2_sweep_g16.log
WL_CLASSES=pcg_mult_dot,pcg_update,pcg_direction,smooth,prolongate,residual python tools/sweep.py 256 16 4 8 16 > gpurun_out/r2_sweep_g16_256.log 2>&1; cat gpurun_out/r2_sweep_g16_256.log
