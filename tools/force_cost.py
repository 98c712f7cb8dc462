import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from waterlily_amd import sim as S
sim = bench.sphere((512,)*3, np.float32)
for _ in range(2): S.sim_step(sim, remeasure=False)
for name, fn in (("pressure_force", lambda: S.pressure_force(sim)), ("viscous_force", lambda: S.viscous_force(sim)), ("total_force", lambda: S.total_force(sim)), ("pressure_moment", lambda: S.pressure_moment((255.,255.,255.), sim))):
    for rep in range(3):
        torch.cuda.synchronize(); t0=time.perf_counter(); f=fn(); torch.cuda.synchronize()
        print(name, rep, round((time.perf_counter()-t0)*1e3,2), "ms", np.round(np.asarray(f),3))
