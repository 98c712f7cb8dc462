#!/usr/bin/env python3
"""where do the oracle and the HIP path part on the accelerating-circle run of tools/longparity.py (moving:64)?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import geometry as G
from oracle import wl_oracle as O
from waterlily_amd import body as B
from waterlily_amd import sim as S
T = np.float64
m = 64; dims = (2 * m, m); radius = m / 8
U = (0.0, 0.0)
so = O.Simulation(dims, U, radius, U=1.0, nu=radius / 250, body=G.Body(G.Sphere((m / 2, m / 2), radius), G.Translate(v=(0.5, 0.0), a=(0.02, 0.0))), T=T)
sh = S.Simulation(dims, U, radius, U=1.0, nu=radius / 250, body=B.Sphere((m / 2, m / 2), radius, 2, map=B.translation(2, v=(0.5, 0.0), a=(0.02, 0.0))), T=T)
first, last = int(sys.argv[1]), int(sys.argv[2])
def mx(a, b): return np.abs(S.to_host(a).astype(np.float64) - b).max()
for k in range(1, last + 1):
    if k < first:
        O.sim_step(so); S.sim_step(sh); continue
    t = float(np.sum(np.asarray(so.flow.dt, dtype=np.float64)))
    th = float(np.sum(np.asarray(sh.flow.dt, dtype=np.float64)))
    O.measure(so); S.measure(sh)
    xc = m / 2 + 0.5 * t + 0.02 * t * t
    g = " ".join(f"{nm} {mx(getattr(sh.flow, nm), getattr(so.flow, nm)):.1e}" for nm in ("mu0", "mu1", "V"))
    dl = " ".join(f"{mx(b.D, a.D):.0e}" for a, b in zip(so.pois.levels, sh.pois.levels))
    O.mom_step(so.flow, so.pois); S.mom_step(sh.flow, sh.pois)
    print(f"{k:4d} t {t:.4f} dt-diff {abs(t - th):.1e} x_c {xc:7.2f} | {g} | D levels {dl} | n {so.pois.n[-2:]} {sh.pois.n[-2:]} du {mx(sh.flow.u, so.flow.u):.2e} dp {mx(sh.flow.p, so.flow.p):.2e} "
          f"umax {np.abs(so.flow.u).max():.2f} dt {so.flow.dt[-1]:.4f} {sh.flow.dt[-1]:.4f}", flush=True)
