#!/usr/bin/env python3
"""ms per sim_step! of small configurations on the HIP path (BASELINE C1: 2-D circle 192x64 Float64; 3-D 64^3 / 128^3 Float32):
the launch-latency end of the size range.  usage: c1_time.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waterlily_amd import body as B, sim as S
for dims, T, Re in (((192, 64), np.float64, 100.0), ((768, 256), np.float32, 100.0), ((64, 64, 64), np.float32, 3700.0), ((128, 128, 128), np.float32, 3700.0), ((256, 256, 256), np.float32, 3700.0), ((128, 128, 128), np.float64, 3700.0)):
    D = len(dims); m = dims[-1]; R, c = m / 8, m / 2 - 1
    U = (1.0,) + (0.0,) * (D - 1)
    s = S.Simulation(dims, U, 2 * R, nu=2 * R / Re, body=B.Sphere((c,) * D, R, D), T=T)
    for _ in range(20):
        S.sim_step(s, remeasure=False)
    torch.cuda.synchronize(); t0 = time.perf_counter(); K = 200
    for _ in range(K):
        S.sim_step(s, remeasure=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    S.set_option(31, 0)                    # the bottom of the V-cycle through global memory (the form before wl_set_option(31))
    for _ in range(5):
        S.sim_step(s, remeasure=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        S.sim_step(s, remeasure=False)
    torch.cuda.synchronize(); dt0 = (time.perf_counter() - t0) / K
    S.set_option(31, 1)
    print(f"{dims} {np.dtype(T).name}: {dt * 1e3:.3f} ms/step = {np.prod(dims) / dt / 1e6:.1f} MLUPS, V-cycles {s.pois.n[-2:]}   (option 31 off: {dt0 * 1e3:.3f} ms)", flush=True)
