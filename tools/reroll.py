#!/usr/bin/env python3
"""Does the step time depend on where the fields landed in HBM, and can it be re-rolled inside one process?
Builds the 512^3 sphere case `rolls` times one after the other (each time freeing everything and emptying torch's cache
first) and times 5 steady steps of each.  usage: reroll.py [size] [rolls] [hold]   (hold=1: keep a 4 GB spacer between rolls)"""
import gc, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from waterlily_amd import sim as S
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rolls = int(sys.argv[2]) if len(sys.argv) > 2 else 5
hold = len(sys.argv) > 3 and sys.argv[3] == "1"
spacers = []
for r in range(rolls):
    sim = bench.sphere((size,) * 3, np.float32)
    for _ in range(8):
        S.sim_step(sim, remeasure=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        S.sim_step(sim, remeasure=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"roll {r}: {ms:.2f} ms/step  p at 0x{sim.flow.p.data_ptr():x}  u at 0x{sim.flow.u.data_ptr():x}", flush=True)
    del sim
    gc.collect(); torch.cuda.empty_cache()
    if hold:
        spacers.append(torch.empty((1 << 30) + 12345 * (r + 1), dtype=torch.float32, device="cuda"))
