#!/usr/bin/env python3
"""Where measure!(sim) spends its time at 512^3 (device geometry): torch profiler table + section timings."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from waterlily_amd import sim as S
from waterlily_amd import body as B
m = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sim = bench.sphere((m,) * 3, np.float32)
a = sim.flow
def tm(name, fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print(f"{name:34s} {(time.perf_counter() - t0) / n * 1e3:8.2f} ms")
dims = tuple(n - 2 for n in a.N)
tm("measure!(sim) total", lambda: S.measure(sim))
tm("  measure_fields_into", lambda: B.measure_fields_into(sim.body, dims, a.mu0, a.mu1, a.V, a.sigma, t=0.0, eps=sim.eps, slab=a.slab))
tm("  fills only", lambda: (B._fill_whole(a.mu0, 1), B._fill_whole(a.mu1, 0), B._fill_whole(a.V, 0)))
tm("  BC mu0 + BC V", lambda: (S.BC(a.mu0, (0.0,) * 3, False, a.perdir), S.BC(a.V, (0.0,) * 3, a.exitBC, a.perdir)))
tm("  flow_update (row flags)", lambda: S.flow_update(a))
tm("  update!(pois)", lambda: S.update(sim.pois))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    B.measure_fields_into(sim.body, dims, a.mu0, a.mu1, a.V, a.sigma, t=0.0, eps=sim.eps, slab=a.slab)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
