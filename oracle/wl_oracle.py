"""ctypes front-end of the CPU oracle (oracle/wl_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by bench.py's
``cpu_baseline`` leg -- never by the product package ``waterlily_amd``.

The classes mirror the reference's Julia API (names, argument meaning, error behaviour) so that the
pin tests in tests/test_oracle_pins.py read like /root/reference/test/maintests.jl:

    Flow              src/Flow.jl:92-122          MultiLevelPoisson  src/MultiLevelPoisson.jl:44-60
    Poisson           src/Poisson.jl:21-38        Simulation         src/WaterLily.jl:59-79
    mom_step / sim_step / sim_time / measure      src/Flow.jl:153-169, src/WaterLily.jl:89-119

Arrays are numpy, Fortran order, shaped like the Julia arrays ((N1,N2[,N3]) scalars, (...,D) vectors,
(...,D,D) for mu1), one ghost layer per side.  Indices and component numbers are 0-based in Python.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Callable, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Grid(C.Structure):
    _fields_ = [("D", C.c_int), ("n", C.c_int * 3), ("s", C.c_long * 3), ("ncell", C.c_long)]

    @staticmethod
    def of(N: Sequence[int]) -> "Grid":
        D = len(N)
        g = Grid()
        g.D = D
        n = list(N) + [1] * (3 - D)
        g.n[:] = n
        g.s[:] = [1, n[0], n[0] * n[1]]
        g.ncell = n[0] * n[1] * n[2]
        return g


def build(force: bool = False) -> str:
    """Compile oracle/libwlo.so with the committed Makefile (gcc, OpenMP)."""
    so = os.path.join(_HERE, "libwlo.so")
    srcs = [os.path.join(_HERE, f) for f in ("wl_oracle.c", "wlo_impl.h", "Makefile")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B", "libwlo.so"], check=True, capture_output=True)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _declare(_LIB)
    return _LIB


def _suf(T) -> str:
    T = np.dtype(T)
    if T == np.float32:
        return "_f32"
    if T == np.float64:
        return "_f64"
    raise TypeError(f"unsupported element type {T}")


def _declare(L: C.CDLL) -> None:
    vp, gp, dp, i, d, lg = C.c_void_p, C.POINTER(Grid), C.POINTER(C.c_double), C.c_int, C.c_double, C.c_long
    for s in ("_f32", "_f64"):
        fl = C.c_float if s == "_f32" else C.c_double

        def f(name, res, *args):
            fn = getattr(L, name + s)
            fn.restype = res
            fn.argtypes = list(args)

        f("wlo_dot", fl, vp, vp, lg)
        f("wlo_L2_inside", d, vp, gp)
        f("wlo_bc_vec", None, vp, gp, dp, i, i)
        f("wlo_bc_per", None, vp, gp, i)
        f("wlo_exit_bc", None, vp, vp, gp, dp, d)
        f("wlo_conv_diff", None, vp, vp, vp, gp, d, i)
        f("wlo_accelerate", None, vp, gp, dp)
        f("wlo_bdim", None, vp, vp, vp, vp, vp, vp, gp, d)
        f("wlo_scale_u", None, vp, gp, d)
        f("wlo_div", None, vp, vp, gp)
        f("wlo_cfl", d, vp, vp, gp, d)
        f("wlo_set_diag", None, vp, vp, vp, gp)
        f("wlo_mult", None, vp, vp)
        f("wlo_residual", None, vp)
        f("wlo_increment", None, vp)
        f("wlo_jacobi", None, vp, i)
        f("wlo_pcg", i, vp, i)
        f("wlo_L2", fl, vp)
        f("wlo_solver", i, vp, d, i)
        f("wlo_restrictL", None, vp, gp, vp, gp, i)
        f("wlo_restrict", None, vp, gp, vp, gp)
        f("wlo_prolongate", None, vp, gp, vp, gp)
        f("wlo_mg_create", vp, vp, vp, vp, gp, i, i)
        f("wlo_mg_destroy", None, vp)
        f("wlo_mg_update", None, vp)
        f("wlo_vcycle", None, vp, i)
        f("wlo_mg_solver", i, vp, d, i)
        f("wlo_mg_nlevels", i, vp)
        f("wlo_mg_array", vp, vp, i, i, C.POINTER(C.c_int))
        f("wlo_mg_level", vp, vp, i)
        f("wlo_poisson_create", vp, vp, vp, vp, gp, i)
        f("wlo_poisson_destroy", None, vp)
        f("wlo_poisson_array", vp, vp, i)
        f("wlo_project", i, vp, vp, d, d, d, i)
        f("wlo_mom_step", d, vp, vp, d, dp, dp, dp, C.POINTER(C.c_int))
        f("wlo_flow_create", vp, gp, vp, vp, vp, vp, vp, vp, vp, vp, d, i, i)
        f("wlo_flow_destroy", None, vp)
        f("wlo_pforce", None, vp, vp, gp, C.POINTER(C.c_long), dp, lg, dp)
        f("wlo_metric", None, i, vp, vp, gp, i, dp, dp)
        f("wlo_vforce", None, vp, vp, gp, C.POINTER(C.c_long), dp, lg, d, dp)
        f("wlo_pmoment", None, vp, vp, gp, C.POINTER(C.c_long), dp, lg, dp, dp)
        for nm in ("quick", "vanleer"):
            f("wlo_t_" + nm, d, d, d, d)
        f("wlo_t_phi", d, vp, lg)
        f("wlo_t_phiu", d, vp, lg, d)
        f("wlo_t_phiuP", d, vp, lg, lg, d)
        f("wlo_t_phiuL", d, vp, lg, d)
        f("wlo_t_phiuR", d, vp, lg, d)


def _fn(name: str, T):
    return getattr(lib(), name + _suf(T))


def _p(a: np.ndarray):
    assert a.flags.f_contiguous, "oracle arrays must be Fortran-contiguous"
    return a.ctypes.data_as(C.c_void_p)


def _d3(v) -> "C.Array":
    v = list(np.asarray(v, dtype=np.float64).ravel()) + [0.0] * 3
    return (C.c_double * 3)(*v[:3])


def permask(perdir: Sequence[int]) -> int:
    """perdir holds 0-based periodic directions (the reference's are 1-based)."""
    m = 0
    for j in perdir:
        m |= 1 << int(j)
    return m


def zeros(shape, T) -> np.ndarray:
    return np.zeros(tuple(shape), dtype=T, order="F")


# --------------------------------------------------------------------------- util.jl helpers

def loc(i: int, shape: Sequence[int], T=np.float64) -> np.ndarray:
    """util.jl:160  loc(i,I) for every I of an array of extents `shape`; i=-1 is the cell centre
    (the reference's i=0).  Returns an array (D, *shape)."""
    D = len(shape)
    ax = [np.arange(n, dtype=np.float64) - 0.5 for n in shape]  # (I-1.5) with I = q+1
    x = np.stack(np.meshgrid(*ax, indexing="ij"))
    if i >= 0:
        x[i] -= 0.5
    return np.asfortranarray(x.astype(T)) if T != np.float64 else x


def apply_vec(f: Callable, c: np.ndarray) -> None:
    """util.jl:171 applyV!: c[I,i] = f(i, loc(i,I)) on every face of a staggered vector array."""
    D = c.ndim - 1
    for i in range(D):
        x = loc(i, c.shape[:-1])
        c[..., i] = np.broadcast_to(np.asarray(f(i, x), dtype=np.float64), c.shape[:-1]).astype(c.dtype)


def apply_scalar(f: Callable, c: np.ndarray) -> None:
    """util.jl:172 applyS!: c[I] = f(loc(0,I))."""
    x = loc(-1, c.shape)
    c[...] = np.broadcast_to(np.asarray(f(x), dtype=np.float64), c.shape).astype(c.dtype)


def inside(a: np.ndarray):
    return tuple(slice(1, n - 1) for n in a.shape)


def L2(a: np.ndarray) -> float:
    """util.jl:68"""
    return float(np.sum(np.asarray(a[inside(a)], dtype=np.float64) ** 2))


def BC(a: np.ndarray, A, saveexit: bool = False, perdir: Sequence[int] = ()) -> None:
    """util.jl:192-210"""
    g = Grid.of(a.shape[:-1])
    _fn("wlo_bc_vec", a.dtype)(_p(a), C.byref(g), _d3(A), int(saveexit), permask(perdir))


def perBC(a: np.ndarray, perdir: Sequence[int]) -> None:
    """util.jl:227-231"""
    g = Grid.of(a.shape)
    _fn("wlo_bc_per", a.dtype)(_p(a), C.byref(g), permask(perdir))


def exitBC(u: np.ndarray, u0: np.ndarray, U, dt: float) -> None:
    """util.jl:216-222"""
    g = Grid.of(u.shape[:-1])
    _fn("wlo_exit_bc", u.dtype)(_p(u), _p(u0), C.byref(g), _d3(U), float(dt))


_exit_bc = exitBC  # Flow.__init__ has a keyword of the same name


def BCTuple(U, dt: Sequence[float], D: int):
    """Flow.jl:79-80: U may be a tuple or a function (i,t) evaluated at t=sum(dt)."""
    if callable(U):
        t = float(np.sum(np.asarray(dt, dtype=np.float64)))
        return tuple(float(U(i, t)) for i in range(D))
    return tuple(float(x) for x in U)


def _dUdt(U: Callable, i: int, t: float) -> float:
    # the reference uses ForwardDiff.derivative (Flow.jl:71-72); host-side scalar, central difference
    h = 1e-5 * max(1.0, abs(t))
    return (float(U(i, t + h)) - float(U(i, t - h))) / (2 * h)


def accel_tuple(g, U, dt: Sequence[float], D: int):
    """Flow.jl:68-73: uniform acceleration g(i,t)+dU_i/dt at t=sum(dt); None when a no-op."""
    if g is None and not callable(U):
        return None
    t = float(np.sum(np.asarray(dt, dtype=np.float64)))
    out = []
    for i in range(D):
        a = 0.0
        if g is not None:
            a += float(g(i, t))
        if callable(U):
            a += _dUdt(U, i, t)
        out.append(a)
    return tuple(out)


# --------------------------------------------------------------------------- Flow.jl

class Flow:
    """src/Flow.jl:92-122"""

    def __init__(self, N, U, *, dt=0.25, nu=0.0, g=None, ulam=None, perdir=(), exitBC=False, T=np.float64):
        D = len(N)
        self.D, self.T = D, np.dtype(T)
        Ng = tuple(int(n) + 2 for n in N)
        self.N = Ng
        self.grid = Grid.of(Ng)
        self.U, self.g, self.nu = U, g, float(nu)
        self.perdir, self.exitBC = tuple(perdir), bool(exitBC)
        self.dt = [float(np.dtype(T).type(dt))]
        self.u = zeros(Ng + (D,), T)
        apply_vec(ulam if ulam is not None else (lambda i, x: 0.0), self.u)
        U0 = BCTuple(U, [0.0], D)
        BC(self.u, U0, exitBC, perdir)
        _exit_bc(self.u, self.u, U0, 0.0)
        self.u0 = self.u.copy(order="F")
        self.f = zeros(Ng + (D,), T)
        self.p = zeros(Ng, T)
        self.sigma = zeros(Ng, T)
        self.V = zeros(Ng + (D,), T)
        self.mu0 = np.ones(Ng + (D,), dtype=T, order="F")
        self.mu1 = zeros(Ng + (D, D), T)
        BC(self.mu0, (0.0,) * D, False, perdir)
        self._h = _fn("wlo_flow_create", T)(
            C.byref(self.grid), _p(self.u), _p(self.u0), _p(self.f), _p(self.p), _p(self.sigma), _p(self.V),
            _p(self.mu0), _p(self.mu1), self.nu, int(self.exitBC), permask(self.perdir))

    def __del__(self):
        try:
            _fn("wlo_flow_destroy", self.T)(self._h)
        except Exception:
            pass


def time(a: Flow) -> float:
    """Flow.jl:129"""
    return float(np.sum(np.asarray(a.dt[:-1], dtype=np.float64)))


def conv_diff(r, u, Phi, nu=0.1, perdir=()):
    """Flow.jl:36-51"""
    g = Grid.of(u.shape[:-1])
    _fn("wlo_conv_diff", u.dtype)(_p(r), _p(u), _p(Phi), C.byref(g), float(nu), permask(perdir))


def accelerate(r, acc):
    g = Grid.of(r.shape[:-1])
    _fn("wlo_accelerate", r.dtype)(_p(r), C.byref(g), _d3(acc))


def BDIM(a: Flow):
    """Flow.jl:131-135"""
    _fn("wlo_bdim", a.T)(_p(a.u), _p(a.u0), _p(a.f), _p(a.V), _p(a.mu0), _p(a.mu1), C.byref(a.grid), a.dt[-1])


def CFL(a: Flow) -> float:
    """Flow.jl:172-175"""
    return float(_fn("wlo_cfl", a.T)(_p(a.sigma), _p(a.u), C.byref(a.grid), a.nu))


# --------------------------------------------------------------------------- Poisson.jl / MultiLevelPoisson.jl

_WHICH = {"L": 0, "D": 1, "iD": 2, "x": 3, "eps": 4, "r": 5, "z": 6}


def _view(ptr, shape, T) -> np.ndarray:
    n = int(np.prod(shape))
    ct = C.c_float if np.dtype(T) == np.float32 else C.c_double
    buf = C.cast(ptr, C.POINTER(ct * n)).contents
    return np.frombuffer(buf, dtype=T).reshape(shape, order="F")


class _Level:
    def __init__(self, T, handle, shape, arrays):
        self.T, self._h, self.shape = np.dtype(T), handle, shape
        for k, v in arrays.items():
            setattr(self, k, v)


class Poisson:
    """src/Poisson.jl:21-38 (single level)."""

    def __init__(self, x: np.ndarray, L: np.ndarray, z: np.ndarray, perdir=()):
        assert x.shape == z.shape and L.shape == x.shape + (x.ndim,)
        self.T = x.dtype
        self.x, self.L, self.z = x, L, z
        self.grid = Grid.of(x.shape)
        self.perdir = tuple(perdir)
        self.n: list[int] = []
        self._h = _fn("wlo_poisson_create", self.T)(_p(x), _p(L), _p(z), C.byref(self.grid), permask(perdir))
        arr = _fn("wlo_poisson_array", self.T)
        self.D = _view(arr(self._h, 1), x.shape, self.T)
        self.iD = _view(arr(self._h, 2), x.shape, self.T)
        self.eps = _view(arr(self._h, 4), x.shape, self.T)
        self.r = _view(arr(self._h, 5), x.shape, self.T)

    def __del__(self):
        try:
            _fn("wlo_poisson_destroy", self.T)(self._h)
        except Exception:
            pass


class MultiLevelPoisson:
    """src/MultiLevelPoisson.jl:44-60"""

    def __init__(self, x: np.ndarray, L: np.ndarray, z: np.ndarray, maxlevels=10, perdir=()):
        assert x.shape == z.shape and L.shape == x.shape + (x.ndim,)
        self.T = x.dtype
        self.x, self.L, self.z = x, L, z
        self.grid = Grid.of(x.shape)
        self.perdir = tuple(perdir)
        self.n: list[int] = []
        self._h = _fn("wlo_mg_create", self.T)(_p(x), _p(L), _p(z), C.byref(self.grid), permask(perdir), maxlevels)
        nl = _fn("wlo_mg_nlevels", self.T)(self._h)
        if nl <= 2:
            _fn("wlo_mg_destroy", self.T)(self._h)
            self._h = None
            raise AssertionError("MultiLevelPoisson requires size=a2ⁿ, where n>2")
        self.levels = []
        D = x.ndim
        for l in range(nl):
            n3 = (C.c_int * 3)()
            arrs = {}
            for k, w in _WHICH.items():
                ptr = _fn("wlo_mg_array", self.T)(self._h, l, w, n3)
                shp = tuple(n3[:D])
                arrs[k] = _view(ptr, shp + ((D,) if k == "L" else ()), self.T)
            self.levels.append(_Level(self.T, _fn("wlo_mg_level", self.T)(self._h, l), shp, arrs))

    def __del__(self):
        try:
            if self._h:
                _fn("wlo_mg_destroy", self.T)(self._h)
        except Exception:
            pass


def _ph(p):
    """handle of the level-1 Poisson of either solver type"""
    return p.levels[0]._h if isinstance(p, MultiLevelPoisson) else p._h


def update(p):
    """Poisson.jl:46 / MultiLevelPoisson.jl:62-68"""
    if isinstance(p, MultiLevelPoisson):
        _fn("wlo_mg_update", p.T)(p._h)
    else:
        g = p.grid
        _fn("wlo_set_diag", p.T)(_p(p.D), _p(p.iD), _p(p.L), C.byref(g))


def mult(p, x: np.ndarray) -> np.ndarray:
    """Poisson.jl:62-68: fills and returns p.z = A x"""
    assert x.shape == p.z.shape
    _fn("wlo_mult", p.T)(_ph(p), _p(x))
    return p.z


def residual(p):
    _fn("wlo_residual", p.T)(_ph(p))


def increment(p):
    _fn("wlo_increment", p.T)(_ph(p))


def Jacobi(p, it=1):
    _fn("wlo_jacobi", p.T)(_ph(p), it)


def pcg(p, it=6) -> int:
    return _fn("wlo_pcg", p.T)(_ph(p), it)


def L2p(p) -> float:
    """Poisson.jl:146"""
    return float(_fn("wlo_L2", p.T)(_ph(p)))


def Vcycle(ml: MultiLevelPoisson, l=0):
    _fn("wlo_vcycle", ml.T)(ml._h, l)


def solver(p, tol=1e-4, itmx=None):
    """Poisson.jl:162-172 / MultiLevelPoisson.jl:87-99"""
    if isinstance(p, MultiLevelPoisson):
        n = _fn("wlo_mg_solver", p.T)(p._h, tol, 32 if itmx is None else int(itmx))
    else:
        n = _fn("wlo_solver", p.T)(p._h, tol, 1000 if itmx is None else int(itmx))
    p.n.append(n)


def restrict(a, b):
    _fn("wlo_restrict", a.dtype)(_p(a), C.byref(Grid.of(a.shape)), _p(b), C.byref(Grid.of(b.shape)))


def prolongate(a, b):
    _fn("wlo_prolongate", a.dtype)(_p(a), C.byref(Grid.of(a.shape)), _p(b), C.byref(Grid.of(b.shape)))


def restrictL(a, b, perdir=()):
    _fn("wlo_restrictL", a.dtype)(_p(a), C.byref(Grid.of(a.shape[:-1])), _p(b), C.byref(Grid.of(b.shape[:-1])),
                                  permask(perdir))


def project(a: Flow, b: MultiLevelPoisson, w=1.0) -> int:
    n = _fn("wlo_project", a.T)(a._h, b._h, a.dt[-1], float(w), 1e-4, 32)
    b.n.append(n)
    return n


def mom_step(a: Flow, b: MultiLevelPoisson):
    """Flow.jl:153-169"""
    U = BCTuple(a.U, a.dt, a.D)
    gp = accel_tuple(a.g, a.U, a.dt[:-1], a.D)
    gc = accel_tuple(a.g, a.U, a.dt, a.D)
    n2 = (C.c_int * 2)()
    dtn = _fn("wlo_mom_step", a.T)(a._h, b._h, a.dt[-1], _d3(U), None if gp is None else _d3(gp),
                                   None if gc is None else _d3(gc), n2)
    b.n.extend([int(n2[0]), int(n2[1])])
    a.dt.append(float(dtn))


def pressure_force_band(p: np.ndarray, df: np.ndarray, idx: np.ndarray, nds: np.ndarray) -> np.ndarray:
    """Metrics.jl:94-100 with nds handed over as a compact band (idx: linear column-major cell
    indices, nds: (nband, D) Float64)."""
    D = p.ndim
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    nds = np.ascontiguousarray(nds, dtype=np.float64)
    out = (C.c_double * 3)()
    _fn("wlo_pforce", p.dtype)(_p(p), _p(df), C.byref(Grid.of(p.shape)), idx.ctypes.data_as(C.POINTER(C.c_long)),
                               nds.ctypes.data_as(C.POINTER(C.c_double)), len(idx), out)
    return np.array(out[:D])


_METRIC = {"ke": 0, "curl": 1, "omega_mag": 2, "omega_theta": 3, "lambda2": 4}


def metric(out: np.ndarray, kind: str, u: np.ndarray, i: int = 0, par=None, par2=None) -> np.ndarray:
    """Metrics.jl:14-77 over inside(out)"""
    _fn("wlo_metric", u.dtype)(_METRIC[kind], _p(out), _p(u), C.byref(Grid.of(out.shape)), int(i),
                               None if par is None else _d3(par), None if par2 is None else _d3(par2))
    return out


def viscous_force_band(u: np.ndarray, nu: float, df: np.ndarray, idx: np.ndarray, nds: np.ndarray) -> np.ndarray:
    """Metrics.jl:109-113 with the compact nds band"""
    D = u.ndim - 1
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    nds = np.ascontiguousarray(nds, dtype=np.float64)
    out = (C.c_double * 3)()
    _fn("wlo_vforce", u.dtype)(_p(u), _p(df), C.byref(Grid.of(u.shape[:-1])), idx.ctypes.data_as(C.POINTER(C.c_long)),
                               nds.ctypes.data_as(C.POINTER(C.c_double)), len(idx), float(nu), out)
    return np.array(out[:D])


def pressure_moment_band(x0, p: np.ndarray, df: np.ndarray, idx: np.ndarray, nds: np.ndarray) -> np.ndarray:
    """Metrics.jl:130-134 with the compact nds band"""
    D = p.ndim
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    nds = np.ascontiguousarray(nds, dtype=np.float64)
    out = (C.c_double * 3)()
    _fn("wlo_pmoment", p.dtype)(_p(p), _p(df), C.byref(Grid.of(p.shape)), idx.ctypes.data_as(C.POINTER(C.c_long)),
                                nds.ctypes.data_as(C.POINTER(C.c_double)), len(idx), _d3(x0), out)
    return np.array(out[:D])


# --------------------------------------------------------------------------- WaterLily.jl

class Simulation:
    """src/WaterLily.jl:59-79.  `body`: an oracle.geometry.Body (closed-form sdf/map: measured by oracle.geometry, the
    default) or any object together with `measure_fn(body, dims, t=, eps=, T=)` returning the (mu0, mu1, V, d) host
    arrays of Body.jl:32-50 before boundary conditions and `nds_fn(body, dims, t=)` (Metrics.jl:84-87)."""

    def __init__(self, dims, u_BC, L, *, dt=0.25, nu=0.0, g=None, U=None, eps=1, perdir=(), ulam=None,
                 exitBC=False, body=None, T=np.float32, measure_fn=None, nds_fn=None):
        assert not (callable(u_BC) and callable(ulam)), "`u_BC` and `uλ` cannot be both specified as Function"
        assert not (U is None and callable(u_BC)), "`U` must be specified if `u_BC` is a Function"
        D = len(dims)
        if ulam is None:
            ulam = (lambda i, x: u_BC(i, 0.0)) if callable(u_BC) else (lambda i, x: u_BC[i])
        self.U = float(np.sqrt(sum(float(v) ** 2 for v in u_BC))) if U is None else U
        self.L, self.eps = L, eps
        if body is not None and measure_fn is None:
            from . import geometry as _G
            assert isinstance(body, (_G.Body, _G.Bodies)), "give an oracle.geometry.Body / Bodies, or measure_fn/nds_fn for any other body"
            measure_fn, nds_fn = _G.measure_fields, _G.nds_band
        self.body, self._measure_fn, self._nds_fn = body, measure_fn, nds_fn
        self.flow = Flow(dims, u_BC, ulam=ulam, dt=dt, nu=nu, g=g, T=T, perdir=perdir, exitBC=exitBC)
        if body is not None:
            measure_flow(self.flow, body, measure_fn, t=0.0, eps=eps)
        self.pois = MultiLevelPoisson(self.flow.p, self.flow.mu0, self.flow.sigma, perdir=perdir)


def measure_flow(a: Flow, body, measure_fn, t=0.0, eps=1):
    """Body.jl:31-53: fill mu0, mu1, V (host evaluation of the user's sdf/map), then BC!."""
    mu0, mu1, V, d = measure_fn(body, tuple(n - 2 for n in a.N), t=t, eps=eps, T=a.T)
    a.mu0[...] = mu0
    a.mu1[...] = mu1
    a.V[...] = V
    a.sigma[inside(a.sigma)] = d[inside(d)]
    BC(a.mu0, (0.0,) * a.D, False, a.perdir)
    BC(a.V, (0.0,) * a.D, a.exitBC, a.perdir)


def sim_time(sim: Simulation) -> float:
    """WaterLily.jl:89"""
    return time(sim.flow) * sim.U / sim.L


def measure(sim: Simulation, t=None):
    """WaterLily.jl:116-119"""
    t = float(np.sum(np.asarray(sim.flow.dt, dtype=np.float64))) if t is None else t
    measure_flow(sim.flow, sim.body, sim._measure_fn, t=t, eps=sim.eps)
    update(sim.pois)


def sim_step(sim: Simulation, t_end=None, *, remeasure=True, max_steps=None, verbose=False):
    """WaterLily.jl:98-109"""
    if t_end is None:
        if remeasure and sim.body is not None:
            measure(sim)
        mom_step(sim.flow, sim.pois)
        return
    steps0 = len(sim.flow.dt)
    while sim_time(sim) < t_end and (max_steps is None or len(sim.flow.dt) - steps0 < max_steps):
        sim_step(sim, remeasure=remeasure)
        if verbose:
            print(f"tU/L={sim_time(sim):.4f}, Δt={sim.flow.dt[-1]:.3f}")


def pressure_force(sim: Simulation) -> np.ndarray:
    """Metrics.jl:94-95"""
    idx, nds = sim._nds_fn(sim.body, tuple(n - 2 for n in sim.flow.N), t=time(sim.flow))
    return pressure_force_band(sim.flow.p, sim.flow.f, idx, nds)


def viscous_force(sim: Simulation) -> np.ndarray:
    idx, nds = sim._nds_fn(sim.body, tuple(n - 2 for n in sim.flow.N), t=time(sim.flow))
    return viscous_force_band(sim.flow.u, sim.flow.nu, sim.flow.f, idx, nds)


def total_force(sim: Simulation) -> np.ndarray:
    return pressure_force(sim) + viscous_force(sim)


def pressure_moment(x0, sim: Simulation) -> np.ndarray:
    idx, nds = sim._nds_fn(sim.body, tuple(n - 2 for n in sim.flow.N), t=time(sim.flow))
    return pressure_moment_band(x0, sim.flow.p, sim.flow.f, idx, nds)


# known-answer access to the scalar stencil helpers (Flow.jl:3-9)
def quick(u, c, d, T=np.float64):
    return _fn("wlo_t_quick", T)(u, c, d)


def vanLeer(u, c, d, T=np.float64):
    return _fn("wlo_t_vanleer", T)(u, c, d)


def _f1(f, T):
    return np.asfortranarray(np.asarray(f, dtype=T))


def phi(I, f, T=np.float64):
    f = _f1(f, T)
    return _fn("wlo_t_phi", T)(_p(f), I)


def phiu(I, f, u, T=np.float64):
    f = _f1(f, T)
    return _fn("wlo_t_phiu", T)(_p(f), I, u)


def phiuP(Ip, I, f, u, T=np.float64):
    f = _f1(f, T)
    return _fn("wlo_t_phiuP", T)(_p(f), Ip, I, u)


def phiuL(I, f, u, T=np.float64):
    f = _f1(f, T)
    return _fn("wlo_t_phiuL", T)(_p(f), I, u)


def phiuR(I, f, u, T=np.float64):
    f = _f1(f, T)
    return _fn("wlo_t_phiuR", T)(_p(f), I, u)
