#!/usr/bin/env python3
"""V-cycles per solve of the moving-cylinder case (bench.py --body cylinder) for the first steps, with the solver log of one
step: shows where Float32 stalls at its rounding floor (L2 of the residual cannot fall below tol = 1e-4 once |x| * eps * sqrt(cells)
exceeds it) and Float64 does not.  usage: cyl_cycles.py <size> <f32|f64> [steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from waterlily_amd import sim as S  # noqa: E402

m = int(sys.argv[1])
T = {"f32": np.float32, "f64": np.float64}[sys.argv[2]]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
sim = bench.moving_cylinder((m, m, m), T)
S.solver_log(sim.pois, True)
for k in range(steps):
    n0 = len(sim.pois.n)
    S.sim_step(sim, remeasure=True)
    rows = S.read_solver_log(sim.pois)
    print(f"step {k}: V-cycles {list(sim.pois.n[n0:])} dt {sim.flow.dt[-1]:.6f}")
    if k == 1:
        for n, rinf, r2 in rows:
            print(f"   n={int(n):2d} Linf={rinf:.3e} L2={r2:.3e}")
