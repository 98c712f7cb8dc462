#!/usr/bin/env python3
"""Per-kernel-class time of one sim_step! on the BASELINE sphere, A/B over one wl_set_option key, in ONE process.
usage: classes.py <size> <key> [reps] [dtype] [valA valB]   (default values 1 0)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from waterlily_amd import _lib  # noqa: E402
from waterlily_amd import sim as S  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
key = int(sys.argv[2]) if len(sys.argv) > 2 else 9
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
T = np.float64 if (len(sys.argv) > 4 and sys.argv[4] == "f64") else np.float32
VA, VB = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (1, 0)
L = _lib.lib()
sim = bench.sphere((size,) * 3, T)
names = {L.wl_kernel_name(k).decode(): k for k in range(24)}
for _ in range(int(os.environ.get('WL_PRESTEPS', '30'))):   # past the impulsive start: 1 V-cycle per solve
    S.sim_step(sim, remeasure=False)
classes = ["pcg_mult_dot", "pcg_update", "pcg_direction", "pcg_init", "smooth", "residual", "conv_diff", "bdim", "correct", "div",
           "scale", "cfl"]
res = {}
for r in range(reps):
    for val in (VA, VB):
        _lib.check(L.wl_set_option(key, val))
        for nm in classes:
            if nm not in names:
                continue
            _lib.check(L.wl_prof_reset())
            _lib.check(L.wl_prof_select(names[nm], int(0.5 * size ** 3)))
            S.sim_step(sim, remeasure=False)
            nl, nc, ms = C.c_int64(), C.c_int64(), C.c_double()
            _lib.check(L.wl_prof_timed(C.byref(nl), C.byref(nc), C.byref(ms)))
            res.setdefault((nm, val), []).append((ms.value, nl.value))
        _lib.check(L.wl_prof_select(-1, 0))
_lib.check(L.wl_set_option(key, VA))
print(f"{size}^3 {T.__name__}: per-class ms per step (launches), option[{key}] = {VA} | {VB};  uniform rows level 0: {S.uniform_rows(sim.pois, 0)}")
for nm in classes:
    if (nm, VA) not in res:
        continue
    pl = {}
    for val in (VA, VB):
        ms = sum(x[0] for x in res[(nm, val)]); n = sum(x[1] for x in res[(nm, val)])
        pl[val] = (ms / max(1, n), n / len(res[(nm, val)]))
    print(f"  {nm:14s} {pl[VA][0]:7.3f} | {pl[VB][0]:7.3f} ms per launch   ({pl[VA][1]:.1f} | {pl[VB][1]:.1f} launches per step)")
