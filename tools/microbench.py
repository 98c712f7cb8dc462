#!/usr/bin/env python3
"""In-process A/B timing of hot-path operators on a size^3 Poisson problem (hipEvent timing inside libwlhip).
Variants alternate within ONE process (box-to-box and clock variance cancels).  usage: microbench.py [size] [reps]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waterlily_amd import _lib  # noqa: E402
from waterlily_amd import sim as S  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
L = _lib.lib()
T = np.float32
a = S.Flow((size,) * 3, (1.0, 0.0, 0.0), T=T)
g = torch.Generator(device="cuda").manual_seed(0)
for f in (a.u, a.p, a.sigma):
    f.copy_(torch.rand(f.shape, generator=g, device="cuda", dtype=f.dtype) - 0.5)
a.mu0.copy_(0.5 + 0.5 * torch.rand(a.mu0.shape, generator=g, device="cuda", dtype=a.mu0.dtype))
S.BC(a.mu0, (0, 0, 0))
ml = S.MultiLevelPoisson(a.p, a.mu0, a.sigma)
names = {L.wl_kernel_name(k).decode(): k for k in range(24)}
ncell = size ** 3


def timed(kclass, fn):
    _lib.check(L.wl_prof_reset())
    _lib.check(L.wl_prof_select(names[kclass], int(0.9 * ncell)))
    fn()
    nl, nc, ms = C.c_int64(), C.c_int64(), C.c_double()
    _lib.check(L.wl_prof_timed(C.byref(nl), C.byref(nc), C.byref(ms)))
    _lib.check(L.wl_prof_select(-1, 0))
    return ms.value / max(1, nl.value), nl.value


def ab(title, kclass, key, fn, algT, vals=(1, 0), restore=1):
    res = {v: [] for v in vals}
    for r in range(reps):
        for val in vals:
            _lib.check(L.wl_set_option(key, val))
            S.residual(ml)
            t, n = timed(kclass, fn)
            res[val].append(t)
    _lib.check(L.wl_set_option(key, restore))
    for val in vals:
        t = float(np.median(res[val]))
        print(f"{title:28s} option[{key}]={val}: {t:7.3f} ms/launch  {algT * 4 * ncell / max(t, 1e-9) / 1e6:7.0f} GB/s alg ({n} launches)")


ab("pcg mult+dot (6T)", "pcg_mult_dot", 0, lambda: S.pcg(ml), 6)
ab("increment (9T)", "increment", 0, lambda: S.increment(ml), 9)
ab("residual (8T)", "residual", 0, lambda: S.residual(ml), 8)
ab("V-cycle smoother (9T)", "smooth", 1, lambda: S.Vcycle(ml), 9)
print("-- option[4]: rows per thread of the 7-point kernel (1 | 2)")
ab("pcg mult+dot (6T)", "pcg_mult_dot", 4, lambda: S.pcg(ml), 6, (1, 2), 0)
ab("increment (9T)", "increment", 4, lambda: S.increment(ml), 9, (1, 2), 0)
ab("residual (8T)", "residual", 4, lambda: S.residual(ml), 8, (1, 2), 0)
ab("V-cycle smoother (9T)", "smooth", 4, lambda: S.Vcycle(ml), 9, (1, 2), 0)
for nm in ("pcg_update", "pcg_direction", "pcg_init"):
    t, n = timed(nm, lambda: S.pcg(ml))
    print(f"{nm:28s} {t:7.3f} ms/launch ({n} launches)")

# BDIM! inside mom_step (option 3 = body-free row flags): needs a body
import bench  # noqa: E402
sim = bench.sphere((size,) * 3, T)
for val in (1, 0, 1, 0):
    _lib.check(L.wl_set_option(3, val))
    t, n = timed("bdim", lambda: S.sim_step(sim, remeasure=False))
    print(f"BDIM!#2 in mom_step          option[3]={val}: {t:7.3f} ms/launch ({n} launches)")
_lib.check(L.wl_set_option(3, 1))
for val in (1, 0, 1, 0):
    _lib.check(L.wl_set_option(2, val))
    t, n = timed("conv_diff", lambda: S.sim_step(sim, remeasure=False))
    print(f"conv_diff!+BDIM!#1           option[2]={val}: {t:7.3f} ms/launch ({n} launches)")
_lib.check(L.wl_set_option(2, 1))
