#!/usr/bin/env python3
"""Long-run sanity: N steps of the sphere case; V-cycle histogram, dt range, drag coefficient history, NaN check."""
import os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from waterlily_amd import sim as S
m = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
sim = bench.sphere((m,) * 3, np.float32)
R = m / 8
cd = []
for s in range(n):
    S.sim_step(sim, remeasure=False)
    if (s + 1) % max(1, n // 10) == 0:
        f = S.pressure_force(sim)
        cd.append(round(float(-f[0] / (0.5 * np.pi * R ** 2)), 4))     # U = 1; pressure_force returns +oint p n ds
hist = collections.Counter(sim.pois.n)
u = sim.flow.u
print(f"{m}^3, {n} steps: V-cycles {dict(hist)}  dt [{min(sim.flow.dt):.3f}, {max(sim.flow.dt):.3f}]  t*U/L = {S.sim_time(sim):.2f}")
print("pressure drag coefficient history:", cd)
print("finite:", bool(torch.isfinite(u).all()), " max|u| =", float(u.abs().max()))
