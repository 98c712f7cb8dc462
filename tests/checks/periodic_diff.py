#!/usr/bin/env python3
"""Where do a periodic HIP run and the (faithful) oracle part ways?  Steps the reference's Taylor-Green case on both and prints,
per step, the largest difference of u, p, the shell of sigma, dt and the V-cycle counts; then replays the FIRST pressure solve
of a fresh pair operator by operator (residual!, Vcycle!, pcg!) to find the first operator whose output differs."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import wl_oracle as O  # noqa: E402
from waterlily_amd import sim as S  # noqa: E402

T = np.float64 if (len(sys.argv) < 2 or sys.argv[1] == "f64") else np.float32
Lg = 64
k = 2 * math.pi / Lg
nu = 1 / (k * 1e8)


def tgv(i, xy):
    x, y = xy[0] * k, xy[1] * k
    return -np.sin(x) * np.cos(y) if i == 0 else np.cos(x) * np.sin(y)


def mk():
    kw = dict(U=1, ulam=tgv, nu=nu, T=T, perdir=(0, 1))
    return O.Simulation((Lg, Lg), (0, 0), Lg, **kw), S.Simulation((Lg, Lg), (0, 0), Lg, **kw)


def d(a_dev, h):
    g = S.to_host(a_dev).astype(np.float64)
    return float(np.max(np.abs(g - h)) / max(1e-300, np.max(np.abs(h))))


so, sh = mk()
for step in range(3):
    O.sim_step(so, remeasure=False)
    S.sim_step(sh, remeasure=False)
    sg = so.flow.sigma
    shell = np.ones(sg.shape, bool)
    shell[O.inside(sg)] = False
    sgh = S.to_host(sh.flow.sigma)
    print(f"step {step}: du={d(sh.flow.u, so.flow.u):.2e} dp={d(sh.flow.p, so.flow.p):.2e} dshell={np.max(np.abs(sgh[shell] - sg[shell])):.2e} "
          f"ddt={abs(sh.flow.dt[-1] - so.flow.dt[-1]) / so.flow.dt[-1]:.2e} n={so.pois.n[-2:]} {sh.pois.n[-2:]}")

# ---- the solver alone on identical inputs: a zero-mean source inside, ARBITRARY values in sigma's ghost cells
so, sh = mk()
po, ph = so.pois, sh.pois
rng = np.random.default_rng(3)
z = rng.standard_normal(po.z.shape).astype(T)
z[O.inside(z)] -= z[O.inside(z)].mean()
A, lv = po.levels[0], ph.levels[0]
KEYS = ("x", "r", "eps", "z")


def sync_state():
    for k in KEYS:
        S.upload(getattr(lv, k), getattr(A, k))


def state(tag):
    out = [f"d{k}={np.max(np.abs(S.to_host(getattr(lv, k)).astype(np.float64) - getattr(A, k))):.2e}" for k in KEYS]
    print(f"{tag:>28}: " + " ".join(out))


A.z[...] = z
A.x[...] = 0
sync_state()
O.residual(po)
S.residual(ph, 0)
state("residual!")
sync_state()
O.Vcycle(po)
S.Vcycle(ph)
state("Vcycle!")
sync_state()
no = O.pcg(po)
nh = S.pcg(ph)
state(f"pcg! (updates {no} / {nh})")
sync_state()
print("L2", O.L2p(po), S.L2p(ph))
A.z[...] = z
A.x[...] = 0
sync_state()
O.solver(po)
S.solver(ph)
state(f"solver! n={po.n[-1]}/{ph.n[-1]}")
