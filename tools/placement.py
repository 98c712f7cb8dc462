#!/usr/bin/env python3
"""Does the relative placement of the field arrays in HBM set the speed of the streaming kernels?  One device buffer,
the 512^3 case built in it again and again with a different spacing between consecutive fields (sim.set_arena), the
un-gated kernel classes timed per launch each time -- all in ONE process, so the buffer's own physical placement is
the same for every variant.  usage: placement.py [size]"""
import ctypes as C
import gc
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from waterlily_amd import _lib, sim as S  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L = _lib.lib()
names = {L.wl_kernel_name(k).decode(): k for k in range(24)}
classes = ["smooth", "prolongate", "conv_diff", "correct", "bdim", "cfl"]
MB = 1 << 20
variants = [("torch", None, None), ("packed256", 256, 0), ("2MB", 2 * MB, 0), ("2MB+2MB", 2 * MB, 2 * MB), ("2MB+6MB", 2 * MB, 6 * MB),
            ("2MB+64K", 2 * MB, 64 * 1024), ("2MB+1MB", 2 * MB, MB), ("1GB", 1 << 30, 0), ("1GB+2MB", 1 << 30, 2 * MB),
            ("torch", None, None), ("2MB", 2 * MB, 0)]
need = int((size + 2) ** 3 * 4 * 36 * 1.25)
print(f"{size}^3: ms per finest-level launch;   " + "  ".join(f"{c:>10s}" for c in classes) + "    step")
for tag, rnd, skew in variants:
    if rnd is None:
        S.set_arena(0, "cuda:0")
    else:
        S.set_arena(need * (3 if rnd >= (1 << 30) else 1), "cuda:0", rnd, skew)
    sim = bench.sphere((size,) * 3, np.float32)
    for _ in range(9):
        S.sim_step(sim, remeasure=False)
    row = []
    for nm in classes:
        _lib.check(L.wl_prof_reset())
        _lib.check(L.wl_prof_select(names[nm], int(0.5 * size ** 3)))
        S.sim_step(sim, remeasure=False)
        nl, nc, ms = C.c_int64(), C.c_int64(), C.c_double()
        _lib.check(L.wl_prof_timed(C.byref(nl), C.byref(nc), C.byref(ms)))
        row.append(ms.value / max(1, nl.value))
    _lib.check(L.wl_prof_select(-1, 0))
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4):
        S.sim_step(sim, remeasure=False)
    torch.cuda.synchronize()
    print(f"{tag:>12s} n={sim.pois.n[-2:]}  " + "  ".join(f"{t:10.3f}" for t in row) + f"   {(time.perf_counter() - t0) / 4 * 1e3:7.2f}", flush=True)
    del sim
    gc.collect()
    torch.cuda.synchronize()
