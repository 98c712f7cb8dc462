"""oracle/geometry.py -- TEST INFRASTRUCTURE: CPU restatement of the reference's body measurement for bodies whose
signed-distance function and coordinate map have CLOSED-FORM derivatives.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; it shares no code with waterlily_amd (numpy only, no torch, no
autograd), so it is an independent checker of waterlily_amd.body and of the HIP `measure!` kernel.

What it follows (paths relative to /root/reference):
  * measure!(flow, body; t, eps)     src/Body.jl:31-53   (the fill loop over inside(p), before the two BC! calls)
  * kern, kern0, kern1, mu0, mu1     src/Body.jl:56-61
  * measure(sdf, map, x, t; fastd2)  src/AutoBody.jl:115-131  with ForwardDiff.gradient / jacobian / derivative
                                     replaced by the analytic derivatives of each family below
  * nds(body, x, t)                  src/Metrics.jl:84-87
  * loc(i, I, T)                     src/util.jl:160

Precision.  The reference evaluates `measure(body, loc(i,I,T), t)` with positions of the field type T (Body.jl:37), but
the closures of every BASELINE body carry Float64 constants (`radius, center = m/8, m/2-1`, README.md:41-44,118-120; test
bodies `x .- 2radius`, maintests.jl:371-384), so Julia promotes the whole evaluation to Float64; loc() values are
half-integers, exact in either type.  This module therefore evaluates in Float64 from exact positions and rounds to T
where the reference stores into T arrays (sigma = d at the cell centre, mu0, mu1, V).  The band test `d[I]^2 < (2+eps)^2`
is made on the stored T value, as Body.jl:35 does.

Pinned by the reference's own known answers: tests/test_oracle_geometry.py restates maintests.jl:183-206 (kernel moments,
measure() of `norm2(x)-2-t` and of the `x .+ t^2` map, the fastd2 short cut) on these closed forms.
"""
from __future__ import annotations

import math

import numpy as np

__all__ = ["Sphere", "Cylinder", "Torus", "Plate", "Bodies", "Identity", "Translate", "Rotate2D", "Bend2D", "Body", "measure", "sdf",
           "measure_fields", "nds_band", "mu0", "mu1", "kern"]


# ------------------------------------------------------------------ sdf families: val(xi) -> (M,), grad(xi) -> (D,M)
class Sphere:
    """sqrt(sum(abs2, xi - center)) - radius (circle in 2-D).  `center` scalar or per-axis; `growth`: radius + growth*t
    (the reference's test body `norm2(x)-2-t`, maintests.jl:193)."""

    def __init__(self, center, radius, growth=0.0):
        self.c, self.R, self.g = center, float(radius), float(growth)

    def _e(self, xi):
        c = np.asarray(self.c, dtype=np.float64)
        return xi - (c.reshape(-1, 1) if c.ndim else c)

    def val(self, xi, t):
        return np.sqrt((self._e(xi) ** 2).sum(0)) - self.R - self.g * t

    def grad(self, xi, t):
        e = self._e(xi)
        with np.errstate(invalid="ignore", divide="ignore"):
            return e / np.sqrt((e ** 2).sum(0))          # 0/0 = NaN at the centre, like ForwardDiff


class Cylinder:
    """sqrt(sum over `axes` of (xi - center)^2) - radius: a circle extruded along the other axes (the reference's 3-D
    cylinder examples: `norm2(x[1:2] .- center) - radius`).  Derivatives along the extrusion axes are exactly 0."""

    def __init__(self, center, radius, axes=(0, 1)):
        self.c, self.R, self.axes = center, float(radius), tuple(axes)

    def _e(self, xi):
        c = np.broadcast_to(np.asarray(self.c, dtype=np.float64), (xi.shape[0],))
        e = np.zeros_like(xi)
        for a in self.axes:
            e[a] = xi[a] - c[a]
        return e

    def val(self, xi, t):
        return np.sqrt((self._e(xi) ** 2).sum(0)) - self.R

    def grad(self, xi, t):
        e = self._e(xi)
        with np.errstate(invalid="ignore", divide="ignore"):
            return e / np.sqrt((e ** 2).sum(0))


class Torus:
    """norm((xi1-c1, norm((xi2-c2, xi3-c3)) - R)) - r   (the "donut")"""

    def __init__(self, center, R, r):
        self.c, self.R, self.r = np.broadcast_to(np.asarray(center, dtype=np.float64), (3,)).copy(), float(R), float(r)

    def val(self, xi, t):
        e = xi - self.c[:, None]
        q = np.sqrt(e[1] ** 2 + e[2] ** 2) - self.R
        return np.sqrt(e[0] ** 2 + q ** 2) - self.r

    def grad(self, xi, t):
        e = xi - self.c[:, None]
        s = np.sqrt(e[1] ** 2 + e[2] ** 2)
        q = s - self.R
        rho = np.sqrt(e[0] ** 2 + q ** 2)
        with np.errstate(invalid="ignore", divide="ignore"):
            return np.stack([e[0] / rho, (q / rho) * (e[1] / s), (q / rho) * (e[2] / s)])


class Plate:
    """sqrt(sum(abs2, xi - (clamp(xi1,-a,a), 0))) - thk   (maintests.jl:375, 2-D)"""

    def __init__(self, a, thk):
        self.a, self.thk = float(a), float(thk)

    def _e(self, xi):
        return np.stack([xi[0] - np.clip(xi[0], -self.a, self.a), xi[1]])

    def val(self, xi, t):
        return np.sqrt((self._e(xi) ** 2).sum(0)) - self.thk

    def grad(self, xi, t):
        e = self._e(xi)
        with np.errstate(invalid="ignore", divide="ignore"):
            n = np.sqrt((e ** 2).sum(0))
            # d e1/d xi1 = 1 - clamp' = 0 inside the span, 1 outside
            out = np.abs(xi[0]) > self.a
            return np.stack([np.where(out, e[0] / n, 0.0 * e[0] / n), e[1] / n])


# ------------------------------------------------------------------ maps: xi(x,t), jac(x,t) -> (M,D,D), dot(x,t) -> (D,M)
class Identity:
    identity = True

    def xi(self, x, t):
        return x

    def jac(self, x, t):
        D, M = x.shape
        return np.broadcast_to(np.eye(D), (M, D, D))

    def dot(self, x, t):
        return np.zeros_like(x)


class Translate:
    """xi = x - s(t) with s(t) = s0 + v*t + a*t^2 per axis  (move: v=(1,0); accel: a=(2,0); `x .+ t^2`: a=(-1,-1,..))"""
    identity = False

    def __init__(self, v=0.0, a=0.0, s0=0.0):
        self.v, self.a, self.s0 = (np.asarray(q, dtype=np.float64) for q in (v, a, s0))

    def _col(self, q, D):
        return np.broadcast_to(q, (D,)).reshape(D, 1)

    def xi(self, x, t):
        D = x.shape[0]
        return x - (self._col(self.s0, D) + self._col(self.v, D) * t + self._col(self.a, D) * t * t)

    def jac(self, x, t):
        D, M = x.shape
        return np.broadcast_to(np.eye(D), (M, D, D))

    def dot(self, x, t):
        D = x.shape[0]
        return np.broadcast_to(-(self._col(self.v, D) + 2.0 * self._col(self.a, D) * t), x.shape).copy()


class Rotate2D:
    """xi = R(theta) (x - c), R = [c s; -s c], theta = w*t + th0   (maintests.jl:376-379)"""
    identity = False

    def __init__(self, center, w, th0=0.0):
        self.c, self.w, self.th0 = float(center), float(w), float(th0)

    def _R(self, t):
        s, c = math.sin(self.w * t + self.th0), math.cos(self.w * t + self.th0)
        return np.array([[c, s], [-s, c]]), np.array([[-s, c], [-c, -s]]) * self.w

    def xi(self, x, t):
        return self._R(t)[0] @ (x - self.c)

    def jac(self, x, t):
        return np.broadcast_to(self._R(t)[0], (x.shape[1], 2, 2))

    def dot(self, x, t):
        return self._R(t)[1] @ (x - self.c)


class Bend2D:
    """(x,y) = xy - c; kappa = k1*t + k0; xi = (x + x^3 kappa^2/6, y - x^2 kappa/2)   (maintests.jl:380-383)"""
    identity = False

    def __init__(self, center, k1, k0):
        self.c, self.k1, self.k0 = float(center), float(k1), float(k0)

    def xi(self, x, t):
        X, Y = x[0] - self.c, x[1] - self.c
        k = self.k1 * t + self.k0
        return np.stack([X + X ** 3 * k ** 2 / 6, Y - X ** 2 * k / 2])

    def jac(self, x, t):
        X = x[0] - self.c
        k = self.k1 * t + self.k0
        J = np.zeros((x.shape[1], 2, 2))
        J[:, 0, 0] = 1 + X ** 2 * k ** 2 / 2
        J[:, 1, 0] = -X * k
        J[:, 1, 1] = 1
        return J

    def dot(self, x, t):
        X = x[0] - self.c
        k = self.k1 * t + self.k0
        return np.stack([X ** 3 * k * self.k1 / 3, -X ** 2 * self.k1 / 2])


class Body:
    """AutoBody(sdf, map) (src/AutoBody.jl:13-20, compose=true) for a closed-form family and map"""

    def __init__(self, shape, map=None):
        self.shape, self.map = shape, (map if map is not None else Identity())


class Bodies:
    """Bodies(bodies, ops) (src/AutoBody.jl:40-66): `bodies` are Body objects, ops[i-1] in "+", "-", "&" ("∪", "∩") joins
    bodies[i] to the composite of the ones before it"""

    def __init__(self, bodies, ops=None):
        self.bodies = list(bodies)
        self.ops = ["+"] * (len(self.bodies) - 1) if ops is None else list(ops)
        assert len(self.bodies) == len(self.ops) + 1


def _active(bodies: "Bodies", xp, t):
    """sdf_map_d / reduce_sdf_map (src/AutoBody.jl:73-93): per point the composite distance, the index of the leaf whose
    sdf and map are taken, and the sign of that sdf (-1 after a minus)"""
    d = sdf(bodies.bodies[0], xp, t)
    act = np.zeros(d.shape, dtype=np.int64)
    sgn = np.ones(d.shape)
    for i, (b, op) in enumerate(zip(bodies.bodies[1:], bodies.ops), start=1):
        db = sdf(b, xp, t)
        if op in ("+", "∪", "|"):
            take, new, s = db < d, db, 1.0
        elif op == "-":
            take, new, s = -db > d, -db, -1.0
        else:
            take, new, s = db > d, db, 1.0
        d = np.where(take, new, d)
        act = np.where(take, i, act)
        sgn = np.where(take, s, sgn)
    return d, act, sgn


# ------------------------------------------------------------------ AutoBody.jl:38,115-131
def sdf(body, x, t=0.0):
    if isinstance(body, Bodies):                              # AutoBody.jl:99
        x = np.asarray(x, dtype=np.float64)
        single = x.ndim == 1
        d = _active(body, x[:, None] if single else x, float(t))[0]
        return d[0] if single else d
    x = np.asarray(x, dtype=np.float64)
    single = x.ndim == 1
    xp = x[:, None] if single else x
    d = body.shape.val(body.map.xi(xp, float(t)), float(t))
    return d[0] if single else d


def measure(body, x, t=0.0, fastd2=math.inf):
    """returns (d, n, V); n = V = 0 where d^2 > fastd2 (:118) or where the gradient has a NaN (:120)"""
    if isinstance(body, Bodies):                              # AutoBody.jl:107-110: measure(sdf, map) of the active leaf
        x = np.asarray(x, dtype=np.float64)
        single = x.ndim == 1
        xp = x[:, None] if single else x
        d, act, sgn = _active(body, xp, float(t))
        n, V = np.zeros(xp.shape), np.zeros(xp.shape)
        for i, b in enumerate(body.bodies):
            q = np.nonzero(act == i)[0]
            if not q.size:
                continue
            # the leaf's own measure with its sdf negated after a minus: d -> -d, grad -> -grad (|grad| and V unchanged)
            di, ni, Vi = measure(b, xp[:, q], t, fastd2=fastd2)
            d[q], n[:, q], V[:, q] = sgn[q] * di, sgn[q] * ni, Vi
        if single:
            return d[0], n[:, 0], V[:, 0]
        return d, n, V
    x = np.asarray(x, dtype=np.float64)
    single = x.ndim == 1
    xp = x[:, None] if single else x
    D, M = xp.shape
    t = float(t)
    xi = body.map.xi(xp, t)
    d = body.shape.val(xi, t)
    n = np.zeros((D, M))
    V = np.zeros((D, M))
    near = ~(d * d > fastd2)
    if near.any():
        q = np.nonzero(near)[0]
        J = body.map.jac(xp[:, q], t)                       # (m,D,D)  J[a,b] = d xi_a / d x_b
        g = np.einsum("mab,am->bm", J, body.shape.grad(xi[:, q], t))    # chain rule: grad_x = J^T grad_xi
        ok = ~np.isnan(g).any(0)
        m = np.sqrt((g * g).sum(0))                          # :124
        qq, gg, mm = q[ok], g[:, ok], m[ok]
        d[qq] = d[qq] / mm
        n[:, qq] = gg / mm
        dot = body.map.dot(xp[:, qq], t)                     # :128-130  V = -J \ dot
        V[:, qq] = -np.linalg.solve(J[ok], dot.T[..., None])[..., 0].T
    if single:
        return d[0], n[:, 0], V[:, 0]
    return d, n, V


# ------------------------------------------------------------------ Body.jl:56-61 (Float64)
def kern(d):
    return 0.5 + 0.5 * np.cos(np.pi * d)


def _kern0(d):
    return 0.5 + 0.5 * d + 0.5 * np.sin(np.pi * d) / np.pi


def _kern1(d):
    return 0.25 * (1 - d ** 2) - 0.5 * (d * np.sin(np.pi * d) + (1 + np.cos(np.pi * d)) / np.pi) / np.pi


def mu0(d, eps):
    return _kern0(np.clip(np.asarray(d, dtype=np.float64) / eps, -1, 1))


def mu1(d, eps):
    return eps * _kern1(np.clip(np.asarray(d, dtype=np.float64) / eps, -1, 1))


# ------------------------------------------------------------------ Body.jl:31-50 / Metrics.jl:84-87 over a grid
def _loc(i, idx):
    """util.jl:160 for 0-based index arrays idx (D,M): I.-1.5 (1-based) = idx-0.5; face i additionally -0.5 on axis i"""
    x = idx.astype(np.float64) - 0.5
    if i >= 0:
        x[i] -= 0.5
    return x


def measure_fields(body, dims, t=0.0, eps=1.0, T=np.float32):
    """Body.jl:32-50 (the fill loop, before BC!): returns Fortran-ordered (mu0, mu1, V, d) of the ghosted extents.
    Same signature as the product's host measurement so either can feed oracle.wl_oracle.Simulation."""
    D = len(dims)
    Ng = tuple(int(n) + 2 for n in dims)
    T = np.dtype(T)
    m0 = np.ones(Ng + (D,), T, order="F")
    m1 = np.zeros(Ng + (D, D), T, order="F")
    Vv = np.zeros(Ng + (D,), T, order="F")
    dd = np.zeros(Ng, T, order="F")
    if body is None:
        return m0, m1, Vv, dd
    d2 = T.type((2 + eps) ** 2)
    idx = np.stack(np.meshgrid(*[np.arange(1, n - 1) for n in Ng], indexing="ij")).reshape(D, -1)   # inside(p)
    dc = sdf(body, _loc(-1, idx), t).astype(T)                       # :34   d[I] = sdf(loc(0,I,T)) stored into sigma::T
    dd[tuple(idx)] = dc
    band = (dc * dc) < d2                                            # :35
    ins = (~band) & (dc < 0)                                         # :45
    for i in range(D):
        m0[tuple(idx[:, ins]) + (i,)] = 0
    ib = idx[:, band]
    for i in range(D):
        di, ni, Vi = measure(body, _loc(i, ib), t, fastd2=float((2 + eps) ** 2))   # :37
        Vv[tuple(ib) + (i,)] = Vi[i]
        m0[tuple(ib) + (i,)] = mu0(di, eps)
        k1 = mu1(di, eps)
        for j in range(D):
            m1[tuple(ib) + (i, j)] = k1 * ni[j]
    return m0, m1, Vv, dd


def nds_band(body, dims, t=0.0):
    """Metrics.jl:84-87 over inside(p) in Float64 (Metrics.jl:96): (idx, nds) = column-major linear indices of the cells
    with a non-zero n*kern(clamp(d,-1,1)) and those vectors, sorted by index."""
    D = len(dims)
    Ng = tuple(int(n) + 2 for n in dims)
    if body is None:
        return np.zeros(0, dtype=np.int64), np.zeros((0, D))
    idx = np.stack(np.meshgrid(*[np.arange(1, n - 1) for n in Ng], indexing="ij")).reshape(D, -1)
    d, n, _ = measure(body, _loc(-1, idx), t, fastd2=1.0)
    v = (n * kern(np.clip(d, -1, 1))[None]).T
    keep = (v != 0).any(1)
    lin = np.ravel_multi_index(tuple(idx[:, keep]), Ng, order="F").astype(np.int64)
    order = np.argsort(lin, kind="stable")
    return lin[order], np.ascontiguousarray(v[keep][order])
