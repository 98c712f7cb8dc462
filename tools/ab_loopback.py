#!/usr/bin/env python3
"""In-process A/B of one wl_set_option key on a LOOPBACK rank (bench.py --comm loopback): ms per step with the key at 1 and at 0,
interleaved.  usage: ab_loopback.py <nranks> <key> [reps]   (C4 grid; rank 1)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from waterlily_amd import dist as wd, sim as S  # noqa: E402

P, key = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dims = bench.C4_GRID
wd.init_loopback(min(1, P - 1), P)
nzl = dims[2] // P
sim = bench.sphere(dims, np.float32, fit=(nzl, 2 * nzl))
for _ in range(10):
    S.sim_step(sim, remeasure=False)
torch.cuda.synchronize()
res = {0: [], 1: []}
for r in range(reps):
    for v in (1, 0):
        S.set_option(key, v)
        S.sim_step(sim, remeasure=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            S.sim_step(sim, remeasure=False)
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / 8 * 1e3)
S.set_option(key, 1)
print(f"loopback rank 1 of {P}, C4, key {key}: on {min(res[1]):.3f} ms/step (runs {[round(x, 3) for x in res[1]]}), "
      f"off {min(res[0]):.3f} (runs {[round(x, 3) for x in res[0]]}); V-cycles {sim.pois.n[-4:]}")
