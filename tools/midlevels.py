#!/usr/bin/env python3
"""Time of ONE pcg!(it=6) call on a level of n^3 cells (the multigrid levels below the finest one), as the finest level of
its own small hierarchy: mean over `reps` calls, r restored before each call (the copy is timed separately and
subtracted).  usage: midlevels.py [--f64] [key=v1,v2,...]   e.g.  midlevels.py 15=0,1   (key 24, the partial budget swept in round 3, is a constant now)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from waterlily_amd import _lib  # noqa: E402
from waterlily_amd import sim as S  # noqa: E402

T = np.float64 if "--f64" in sys.argv else np.float32
sweep = [a for a in sys.argv[1:] if "=" in a]
key, vals = (int(sweep[0].split("=")[0]), [int(v) for v in sweep[0].split("=")[1].split(",")]) if sweep else (15, [1])
L = _lib.lib()
reps = 50


def timed(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3     # us


print(f"pcg!(it=6) on an n^3 level, {np.dtype(T).name}: us per call (18 dependent kernels as launched today); option[{key}] sweep")
for n in (256, 128, 64, 32):
    U = (1.0, 0.0, 0.0)
    a = S.Flow((n, n, n), U, T=T)
    R, c = n / 8, n / 2 - 1
    from waterlily_amd.body import AutoBody
    S.measure_flow(a, AutoBody(lambda x, t: torch.sqrt((x[0] - c) ** 2 + (x[1] - c) ** 2 + (x[2] - c) ** 2) - R), 0.0)
    ml = S.MultiLevelPoisson(a.p, a.mu0, a.sigma)
    lv = ml.levels[0]
    g = torch.Generator(device="cuda").manual_seed(1)
    inner = (slice(1, -1),) * 3
    r0 = S.like(lv.r)
    r0[inner] = torch.rand(r0[inner].shape, generator=g, device="cuda", dtype=r0.dtype) - 0.5
    r0[inner] -= r0[inner].mean()
    tcopy = timed(lambda: lv.r.copy_(r0))
    row = []
    for v in vals:
        S.set_option(key, v)

        def call():
            lv.r.copy_(r0)
            _lib.check(L.wl_mg_pcg(ml._h, 0, 6, None))
        row.append(timed(call) - tcopy)
        nu = C.c_int()
        _lib.check(L.wl_mg_pcg(ml._h, 0, 6, C.byref(nu)))
    print(f"  {n:4d}^3: " + "  ".join(f"[{v}] {t:8.1f}" for v, t in zip(vals, row)) + f"   (copy {tcopy:.1f} us, updates of the last call {nu.value})")
    del a, ml
S.set_option(key, 1)
