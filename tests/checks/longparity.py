#!/usr/bin/env python3
"""HIP path and CPU oracle side by side over MANY steps of a BASELINE configuration (README circle / sphere case,
remeasure=false): per step the V-cycle counts of both solves, the relative difference of the time step, max |du| / U,
max |dp| / max |p| and the pressure force of both.  The tests compare a handful of steps; this is the long horizon.
(Lives under tests/: it uses the oracle as the checker; nothing here is product code.)

usage: longparity.py <c1|c2|NxNxN|torus:N|moving:N|pbox:N> <f32|f64> <steps> [every] [self KEY A B]
  pbox:N = a sphere in a 2N x N x N box, periodic in y and z (round 4: the whole-array reductions of the reference -- sigma's
  flux scratch in the ghost cells times eps's periodic copies -- on a long horizon)
  self KEY A B: instead of the oracle, a SECOND HIP simulation stepped with wl_set_option(KEY, B) next to the first with
  wl_set_option(KEY, A) -- e.g. `self 16 16 8` regroups the Float64 partial sums of every dot product (last-bit
  perturbations): how fast does the flow itself amplify rounding differences?"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import geometry as G  # noqa: E402
from oracle import wl_oracle as O  # noqa: E402
from waterlily_amd import body as B  # noqa: E402
from waterlily_amd import sim as S  # noqa: E402

case, tname, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
every = int(sys.argv[4]) if len(sys.argv) > 4 else max(1, steps // 25)
selfcmp = None
if len(sys.argv) > 8 and sys.argv[5] == "self":
    selfcmp = (int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8]))


class _Twin:
    """a second HIP simulation behind the few oracle calls this script makes"""

    def __init__(self, sim):
        self.sim, self.flow, self.pois = sim, self, sim.pois

    dt = property(lambda self: self.sim.flow.dt)
    u = property(lambda self: S.to_host(self.sim.flow.u).astype(np.float64))
    p = property(lambda self: S.to_host(self.sim.flow.p).astype(np.float64))
T = {"f32": np.float32, "f64": np.float64}[tname]
remeasure = False
if case.startswith("torus:"):        # BASELINE C5's case at N^3: torus, Re = 1000
    m = int(case[6:]); dims = (m, m, m); D = 3; Re = 1000.0
    c, Rm, rm = m / 2, m / 4, m / 16
    U = (1.0, 0.0, 0.0)
    so = O.Simulation(dims, U, Rm, nu=Rm / Re, body=G.Body(G.Torus(c, Rm, rm)), T=T)
    sh = S.Simulation(dims, U, Rm, nu=Rm / Re, body=B.Torus((c, c, c), Rm, rm), T=T)
elif case.startswith("pbox:"):
    m = int(case[5:]); dims = (2 * m, m, m); D = 3; Re = 1000.0
    radius, center = m / 8, m / 2 - 1
    U = (1.0, 0.0, 0.0)
    kwp = dict(nu=2 * radius / Re, perdir=(1, 2), T=T)
    so = O.Simulation(dims, U, 2 * radius, body=G.Body(G.Sphere(center, radius)), **kwp)
    sh = S.Simulation(dims, U, 2 * radius, body=B.Sphere((center,) * 3, radius, 3), **kwp)
elif case.startswith("moving:"):     # a circle accelerating through fluid at rest, measure! + update! every step (maintests.jl:391-412 family)
    m = int(case[7:]); dims = (2 * m, m); D = 2; Re = 250.0; remeasure = True
    radius = m / 8
    U = (0.0, 0.0)
    so = O.Simulation(dims, U, radius, U=1.0, nu=radius / Re, body=G.Body(G.Sphere((m / 2, m / 2), radius), G.Translate(v=(0.05, 0.0), a=(5e-6, 0.0))), T=T)
    sh = S.Simulation(dims, U, radius, U=1.0, nu=radius / Re, body=B.Sphere((m / 2, m / 2), radius, 2, map=B.translation(2, v=(0.05, 0.0), a=(5e-6, 0.0))), T=T)
else:
    dims, Re = {"c1": ((192, 64), 100.0), "c2": ((256, 256, 256), 3700.0)}.get(case) or (tuple(int(v) for v in case.split("x")), 3700.0)
    D = len(dims)
    m = dims[-1]
    radius, center = m / 8, m / 2 - 1
    U = (1.0,) + (0.0,) * (D - 1)
    so = O.Simulation(dims, U, 2 * radius, nu=2 * radius / Re, body=G.Body(G.Sphere(center, radius)), T=T)
    sh = S.Simulation(dims, U, 2 * radius, nu=2 * radius / Re, body=B.Sphere((center,) * D, radius, D), T=T)
if selfcmp:
    import copy
    key, va, vb = selfcmp
    body2 = copy.deepcopy(sh.body)
    s2 = S.Simulation(dims, U, sh.L, U=sh.U, nu=sh.flow.nu, body=body2, T=T)
    so = _Twin(s2)
print(f"# {case} {dims} {tname} Re={Re:g}: " + (f"HIP with option[{selfcmp[0]}]={selfcmp[2]} vs HIP with option[{selfcmp[0]}]={selfcmp[1]}" if selfcmp else "oracle (CPU) vs HIP")
      + f", remeasure={str(remeasure).lower()}")
print("# step  V-cycles(oracle)  V-cycles(HIP)  d(dt)/dt   max|du|/U   max|dp|/max|p|   force_x(oracle)  force_x(HIP)   rel")
worst = {"u": 0.0, "p": 0.0, "dt": 0.0, "f": 0.0, "nmis": 0}
for k in range(1, steps + 1):
    if selfcmp:
        S.set_option(key, vb)
        S.sim_step(so.sim, remeasure=remeasure)
        S.set_option(key, va)
    else:
        O.sim_step(so, remeasure=remeasure)
    S.sim_step(sh, remeasure=remeasure)
    no, nh = so.pois.n[-2:], sh.pois.n[-2:]
    ddt = abs(so.flow.dt[-1] - sh.flow.dt[-1]) / so.flow.dt[-1]
    worst["dt"] = max(worst["dt"], ddt)
    worst["nmis"] += int(list(no) != list(nh))
    if k % every == 0 or k == steps or list(no) != list(nh):
        du = np.abs(S.to_host(sh.flow.u).astype(np.float64) - so.flow.u).max()
        pm = max(np.abs(so.flow.p).max(), 1e-300)
        dp = np.abs(S.to_host(sh.flow.p).astype(np.float64) - so.flow.p).max() / pm
        fo, fh = (S.pressure_force(so.sim) if selfcmp else O.pressure_force(so)), S.pressure_force(sh)
        rel = np.abs(fo - fh).max() / max(np.abs(fo).max(), 1e-300)
        worst["u"], worst["p"], worst["f"] = max(worst["u"], du), max(worst["p"], dp), max(worst["f"], rel)
        print(f"{k:6d}  {str(list(no)):>16s}  {str(list(nh)):>13s}  {ddt:9.2e}  {du:10.3e}  {dp:14.3e}  {fo[0]:15.8e}  {fh[0]:14.8e}  {rel:8.1e}", flush=True)
print(f"# worst over the run: max|du|/U {worst['u']:.3e}, max|dp|/max|p| {worst['p']:.3e}, d(dt)/dt {worst['dt']:.2e}, force {worst['f']:.1e}, "
      f"steps with different V-cycle counts: {worst['nmis']} of {steps}")
