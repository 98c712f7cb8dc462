set -e
python -m pytest tests/test_hip_parity.py tests/test_golden.py -q -m gpu -x > gpurun_out/r2_t12.log 2>&1 || { tail -40 gpurun_out/r2_t12.log | cut -c1-600; exit 1; }
tail -3 gpurun_out/r2_t12.log
export WL_PRESTEPS=12
CL=pcg_mult_dot,smooth,prolongate,residual,correct,div,conv_diff
echo "== DPP wave shifts"
WL_CLASSES=$CL python tools/sweep.py 512 4 0 0 > gpurun_out/r2_sweep_dpp.log 2>&1; cat gpurun_out/r2_sweep_dpp.log
cp waterlily_amd/libwlhip.so /tmp/libwlhip_default.so
(cd waterlily_amd/csrc && make -B CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -DWL_NO_DPP" > /dev/null 2>&1)
echo "== ds_bpermute shuffles"
WL_CLASSES=$CL python tools/sweep.py 512 4 0 0 > gpurun_out/r2_sweep_nodpp.log 2>&1; cat gpurun_out/r2_sweep_nodpp.log
cp /tmp/libwlhip_default.so waterlily_amd/libwlhip.so
echo "== DPP again"
WL_CLASSES=$CL python tools/sweep.py 512 4 0 0 > gpurun_out/r2_sweep_dpp2.log 2>&1; cat gpurun_out/r2_sweep_dpp2.log
python tools/longrun.py 256 300 > gpurun_out/r2_longrun.log 2>&1; cat gpurun_out/r2_longrun.log
