import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from waterlily_amd import sim as S
sim = bench.sphere((512,)*3, np.float32)
for _ in range(3): S.sim_step(sim, remeasure=False)
torch.cuda.synchronize()
for rem in (False, True, True):
    t0=time.perf_counter()
    for _ in range(5): S.sim_step(sim, remeasure=rem)
    torch.cuda.synchronize()
    print("remeasure", rem, round((time.perf_counter()-t0)/5*1e3,2), "ms/step")
t0=time.perf_counter()
for _ in range(5): S.measure(sim)
torch.cuda.synchronize()
print("measure alone", round((time.perf_counter()-t0)/5*1e3,2), "ms")
