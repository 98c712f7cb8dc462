#!/usr/bin/env python3
"""In-process A/B of whole sim_step! time for one wl_set_option key on an arbitrary grid.
usage: ab_dims.py nx ny nz key v1 v2 [reps] [--f64]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from waterlily_amd import _lib, sim as S
a = [x for x in sys.argv[1:] if x != "--f64"]
dims, key, vals = tuple(int(v) for v in a[:3]), int(a[3]), (int(a[4]), int(a[5]))
reps = int(a[6]) if len(a) > 6 else 4
T = np.float64 if "--f64" in sys.argv else np.float32
L = _lib.lib()
sim = bench.sphere(dims, T)
for _ in range(6):
    S.sim_step(sim, remeasure=False)
res = {v: [] for v in vals}
for r in range(reps):
    for val in vals:
        _lib.check(L.wl_set_option(key, val))
        S.sim_step(sim, remeasure=False); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            S.sim_step(sim, remeasure=False)
        torch.cuda.synchronize()
        res[val].append((time.perf_counter() - t0) / 3 * 1e3)
for val in vals:
    print(f"{dims} option[{key}]={val}: median {np.median(res[val]):.3f} ms/step  (all: {[round(x, 2) for x in res[val]]})  n={sim.pois.n[-2:]}")
