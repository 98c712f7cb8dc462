#!/usr/bin/env python3
"""In-process A/B of whole sim_step! time for one wl_set_option key.  usage: ab_step.py <size> <key> [reps [valA valB]]
(WL_AB_DTYPE=f64: Float64; WL_AB_LAYOUT=dense: the reference's strides)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from waterlily_amd import _lib, sim as S
size, key = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
vals = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (1, 0)
L = _lib.lib()
sim = bench.sphere((size,) * 3, np.float64 if os.environ.get('WL_AB_DTYPE') == 'f64' else np.float32,
                   padded=os.environ.get('WL_AB_LAYOUT', 'padded') != 'dense')
for _ in range(6):
    S.sim_step(sim, remeasure=False)
res = {v: [] for v in vals}
for r in range(reps):
    for val in vals:
        _lib.check(L.wl_set_option(key, val))
        S.sim_step(sim, remeasure=False); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            S.sim_step(sim, remeasure=False)
        torch.cuda.synchronize()
        res[val].append((time.perf_counter() - t0) / 5 * 1e3)
for val in vals:
    print(f"{size}^3 option[{key}]={val}: median {np.median(res[val]):.3f} ms/step  (all: {[round(x, 2) for x in res[val]]})  n={sim.pois.n[-2:]}")
