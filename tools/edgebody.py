#!/usr/bin/env python3
"""measure! of a body that overlaps the domain boundary: oracle vs product (host closures) vs native kernels"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import geometry as G
from oracle import wl_oracle as O
from waterlily_amd import body as B
from waterlily_amd import sim as S
T = {"f32": np.float32, "f64": np.float64}[sys.argv[1] if len(sys.argv) > 1 else "f64"]
for dims, c0 in (((128, 64), (32.0, 32.0)), ((48, 32, 32), (16.0, 16.0, 16.0))):
    D = len(dims)
    for shift in ([88.0] + [0.0] * (D - 1), [93.5] + [0.0] * (D - 1), [99.0] + [0.0] * (D - 1), [-30.0] + [0.0] * (D - 1), [0.0, 27.0] + [0.0] * (D - 2), [0.0, -29.3] + [0.0] * (D - 2)):
        if D == 3:
            shift = [s * 48 / 128 if i == 0 else s * 0.5 for i, s in enumerate(shift)]
        v = tuple(shift)
        U = (0.0,) * D
        so = O.Simulation(dims, U, 8.0, U=1.0, body=G.Body(G.Sphere(c0, 8.0 if D == 2 else 5.0), G.Translate(v=v)), T=T)
        sn = S.Simulation(dims, U, 8.0, U=1.0, body=B.Sphere(c0, 8.0 if D == 2 else 5.0, D, map=B.translation(D, v=v)), T=T)
        O.measure(so, 1.0)
        S.measure(sn, 1.0)
        out = []
        for k in ("mu0", "mu1", "V"):
            w = getattr(so.flow, k)
            out.append(f"{k} {np.abs(S.to_host(getattr(sn.flow, k)).astype(np.float64) - w).max():.2e}")
        for lvl in range(len(so.pois.levels)):
            a, b = so.pois.levels[lvl], sn.pois.levels[lvl]
            out.append(f"D{lvl} {np.abs(S.to_host(b.D).astype(np.float64) - a.D).max():.1e}")
        print(dims, "shift", [round(x, 2) for x in v], " ".join(out), flush=True)
