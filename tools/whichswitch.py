#!/usr/bin/env python3
"""Which wl_set_option switch changes a bit?  Steps the bench case three times with all switches at their defaults (twice:
run-to-run determinism), then with one switch off at a time, and reports max |du| against the first run.
usage: whichswitch.py <size> [--f64] [--donut]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from waterlily_amd import sim as S  # noqa: E402

n = int(sys.argv[1])
T = np.float64 if "--f64" in sys.argv else np.float32
mk = bench.donut if "--donut" in sys.argv else bench.sphere
KEYS = (3, 8, 9, 13, 14, 18, 19, 22, 23, 30)


def run(off=()):
    for k in off:
        S.set_option(k, 0)
    try:
        sim = mk((n, n, n), T)
        for _ in range(3):
            S.sim_step(sim, remeasure=False)
    finally:
        for k in off:
            S.set_option(k, 1)
    return sim.pois.n[:], list(sim.flow.dt), S.copy_of(sim.flow.u)


base = run()
print("defaults again:", end=" ")
for name, off in [("same", ())] + [(str(k), (k,)) for k in KEYS]:
    r = run(off)
    du = float((r[2] - base[2]).abs().max())
    print(f"off={name:4s} n_equal={r[0] == base[0]} dt_equal={r[1] == base[1]} max|du|={du:.3e}", flush=True)
