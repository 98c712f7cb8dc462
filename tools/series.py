#!/usr/bin/env python3
"""Per-step series of one kernel class's mean finest-level launch time and the step wall time, in one process:
is a slow process slow from its first step to its last (placement of the arrays) or does it speed up (clocks)?
usage: series.py <size> <steps> [class]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from waterlily_amd import _lib, sim as S  # noqa: E402

size, steps = int(sys.argv[1]), int(sys.argv[2])
cls = sys.argv[3] if len(sys.argv) > 3 else "smooth"
L = _lib.lib()
sim = bench.sphere((size,) * 3, np.float32)
names = {L.wl_kernel_name(k).decode(): k for k in range(24)}
out = []
for s in range(steps):
    _lib.check(L.wl_prof_reset())
    _lib.check(L.wl_prof_select(names[cls], int(0.5 * size ** 3)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    S.sim_step(sim, remeasure=False)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    nl, nc, ms = C.c_int64(), C.c_int64(), C.c_double()
    _lib.check(L.wl_prof_timed(C.byref(nl), C.byref(nc), C.byref(ms)))
    out.append((wall, ms.value / max(1, nl.value), sim.pois.n[-2:]))
_lib.check(L.wl_prof_select(-1, 0))
print(f"{size}^3 {cls}: step wall ms / mean launch ms / V-cycles")
for i in range(0, steps, 4):
    print("  ".join(f"{i + q:3d}: {w:6.2f} {m:.3f} {n}" for q, (w, m, n) in enumerate(out[i:i + 4])))
