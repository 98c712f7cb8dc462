#!/usr/bin/env python3
"""Cost of sim_step!(remeasure=true) with a translating sphere (measure! on the device + update!(pois))."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waterlily_amd import sim as S
from waterlily_amd import body as B
from waterlily_amd.body import AutoBody
m = int(sys.argv[1]) if len(sys.argv) > 1 else 512
native = not (len(sys.argv) > 2 and sys.argv[2] == "closures")
R, c = m / 8, m / 2 - 1
sdf = lambda x, t: torch.sqrt((x[0] - c) ** 2 + (x[1] - c) ** 2 + (x[2] - c) ** 2) - R
mp = lambda x, t: x - torch.stack([0.5 * t, torch.zeros_like(t), torch.zeros_like(t)])[:, None]
body = B.Sphere(c, R, 3, map=B.translation(3, v=(0.5, 0.0, 0.0))) if native else AutoBody(sdf, mp)
print("body:", "parametric (HIP measure! kernels)" if native else "closures (torch on the device)")
sim = S.Simulation((m, m, m), (1.0, 0.0, 0.0), 2 * R, nu=2 * R / 3700, body=body, T=np.float32)
for rm in (True, False, True, False):
    S.sim_step(sim, remeasure=rm); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        S.sim_step(sim, remeasure=rm)
    torch.cuda.synchronize()
    print(f"{m}^3 remeasure={rm}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms/step  n={sim.pois.n[-2:]}")
t0 = time.perf_counter(); S.measure(sim); torch.cuda.synchronize(); print(f"measure!(sim) alone: {(time.perf_counter()-t0)*1e3:.1f} ms")
t0 = time.perf_counter(); S.update(sim.pois); torch.cuda.synchronize(); print(f"update!(pois) alone: {(time.perf_counter()-t0)*1e3:.1f} ms")
t0 = time.perf_counter(); f = S.pressure_force(sim); torch.cuda.synchronize(); print(f"pressure_force: {(time.perf_counter()-t0)*1e3:.1f} ms", f)
