#!/usr/bin/env python3
"""Per-launch time of the finest-level kernel classes on the BASELINE sphere for several values of one
wl_set_option key, in ONE process.  usage: sweep.py <size> <key> <v1> <v2> ... [--f64]
env: WL_CLASSES=a,b,..  WL_MINFRAC=0.5 (launches of at least this share of the cells; 0 = every launch of the class)
WL_TOTAL=1 (ms per step summed over the selected launches instead of the per-launch mean)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from waterlily_amd import _lib  # noqa: E402
from waterlily_amd import sim as S  # noqa: E402

argv = [a for a in sys.argv[1:] if a != "--f64"]
T = np.float64 if "--f64" in sys.argv else np.float32
size, key = int(argv[0]), int(argv[1])
vals = [int(v) for v in argv[2:]]
L = _lib.lib()
for kv in os.environ.get("WL_PRESET", "").split(","):   # options that must be set BEFORE the handles are created, "5=1,..."
    if "=" in kv:
        _lib.check(L.wl_set_option(int(kv.split("=")[0]), int(kv.split("=")[1])))
sim = bench.sphere((size,) * 3, T)
names = {L.wl_kernel_name(k).decode(): k for k in range(24)}
for _ in range(int(os.environ.get("WL_PRESTEPS", "30"))):
    S.sim_step(sim, remeasure=False)
classes = os.environ.get("WL_CLASSES", "pcg_mult_dot,pcg_update,pcg_direction,smooth,residual").split(",")
print(f"{size}^3 {T.__name__} option[{key}] sweep; ms per finest-level launch")
print("value  " + "  ".join(f"{c:>13s}" for c in classes))
for rep in range(2):
    for v in vals:
        _lib.check(L.wl_set_option(key, v))
        row = []
        for nm in classes:
            _lib.check(L.wl_prof_reset())
            _lib.check(L.wl_prof_select(names[nm], int(float(os.environ.get("WL_MINFRAC", "0.5")) * size ** 3)))
            S.sim_step(sim, remeasure=False)
            nl, nc, ms = C.c_int64(), C.c_int64(), C.c_double()
            _lib.check(L.wl_prof_timed(C.byref(nl), C.byref(nc), C.byref(ms)))
            row.append(ms.value if os.environ.get("WL_TOTAL") else ms.value / max(1, nl.value))   # WL_TOTAL: ms per step of the class
        _lib.check(L.wl_prof_select(-1, 0))
        import time as _t
        import torch as _torch
        _torch.cuda.synchronize(); t0 = _t.perf_counter()
        for _ in range(3):
            S.sim_step(sim, remeasure=False)
        _torch.cuda.synchronize()
        print(f"{v:5d}  " + "  ".join(f"{t:13.3f}" for t in row) + f"   step {(_t.perf_counter() - t0) / 3 * 1e3:7.2f} ms")
