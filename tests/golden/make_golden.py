#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (run AFTER tests/test_oracle_pins.py is green).

The reference (Julia) cannot run in this environment and ships no golden data, so these vectors are OUTPUTS OF THE
PINNED ORACLE (oracle/wl_oracle.c for the flow and the pressure solver, oracle/geometry.py for the body) on small seeded
inputs -- they are oracle-generated, not reference-generated.  They freeze the oracle's behaviour (a regression guard
for the checker itself) and let the GPU box check the HIP path against committed data.  Inputs are stored next to the
outputs.  No product code (waterlily_amd) takes part in generating them.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import geometry as G  # noqa: E402
from oracle import wl_oracle as O  # noqa: E402


def rnd(shape, T, seed, lo=-1.0, hi=1.0):
    rng = np.random.default_rng(seed)
    return np.asfortranarray((lo + (hi - lo) * rng.random(shape)).astype(T))


def conv_diff_case(Ng, T, perdir, seed):
    D = len(Ng)
    u = rnd(Ng + (D,), T, seed)
    r, Phi = O.zeros(Ng + (D,), T), O.zeros(Ng, T)
    O.conv_diff(r, u, Phi, nu=0.05, perdir=perdir)
    return dict(u=u, r=r, nu=0.05, perdir=np.array(perdir, dtype=np.int64))


def poisson_case(Ng, T, seed):
    D = len(Ng)
    L = rnd(Ng + (D,), T, seed, 0.2, 1.0)
    O.BC(L, (0.0,) * D)
    x, z = rnd(Ng, T, seed + 1), rnd(Ng, T, seed + 2)
    z -= z[O.inside(z)].mean().astype(T)
    x0, z0 = x.copy(order="F"), z.copy(order="F")
    p = O.MultiLevelPoisson(x, L, z)
    O.solver(p)
    return dict(L=L, x0=x0, z0=z0, x=x, n=np.array(p.n), D2=p.levels[1].D.copy(), L2=p.levels[1].L.copy())


def sim_case(dims, T, nsteps, Re):
    m = dims[-1]
    R, c = m / 8, m / 2 - 1
    U = (1.0,) + (0.0,) * (len(dims) - 1)
    s = O.Simulation(dims, U, 2 * R, nu=2 * R / Re, body=G.Body(G.Sphere(c, R)), T=T)
    init = dict(mu0=s.flow.mu0.copy(), mu1=s.flow.mu1.copy(), V=s.flow.V.copy(), u_init=s.flow.u.copy())
    for _ in range(nsteps):
        O.sim_step(s, remeasure=False)
    return dict(u=s.flow.u, p=s.flow.p, n=np.array(s.pois.n), dt=np.array(s.flow.dt), force=O.pressure_force(s),
                dims=np.array(dims), Re=Re, nsteps=nsteps, **init)


def main():
    out = {
        "conv_diff_3d_f32": conv_diff_case((12, 10, 8), np.float32, (), 5),
        "conv_diff_3d_f32_per": conv_diff_case((12, 10, 8), np.float32, (1,), 5),
        "conv_diff_2d_f64": conv_diff_case((18, 12), np.float64, (), 7),
        "poisson_3d_f32": poisson_case((18, 18, 18), np.float32, 20),
        "poisson_2d_f64": poisson_case((34, 18), np.float64, 30),
        "sim_3d_f32": sim_case((16, 16, 16), np.float32, 3, 3700.0),
        "sim_2d_f64": sim_case((32, 16), np.float64, 4, 100.0),
    }
    for name, d in out.items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, {k: getattr(v, "shape", v) for k, v in d.items()})


if __name__ == "__main__":
    main()
