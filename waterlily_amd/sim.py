"""Host-side mirror of the reference's driver API on top of libwlhip.so.

Same names, argument meaning and error behaviour as the reference (file:line relative to /root/reference):

    Flow               src/Flow.jl:92-122            mom_step          src/Flow.jl:153-169
    Poisson            src/Poisson.jl:21-38          MultiLevelPoisson src/MultiLevelPoisson.jl:44-60
    Simulation         src/WaterLily.jl:59-79        sim_step/sim_time/measure  src/WaterLily.jl:89-119
    pressure_force     src/Metrics.jl:94-100

Python-isms: indices, components and periodic directions are 0-based; `!` is dropped from names.
Fields are torch tensors living in HBM (torch only provides device memory and, for multi-GPU,
torch.distributed); they are strided *views* shaped like the Julia arrays ((N1,N2[,N3]) scalars,
(...,D) vectors, (...,D,D) for mu1; one ghost layer) over an allocation whose rows are padded so that the
first interior element of every row is 128-byte aligned.  All numerics run in the HIP library: there is
no CPU fallback and construction raises if no GPU is present.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Optional, Sequence

import numpy as np
import torch

from . import _lib
from . import body as B
from ._lib import FlowDesc, Grid, LevelDesc, check, d3

_TORCH = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}
_WLT = {np.dtype(np.float32): _lib.WL_F32, np.dtype(np.float64): _lib.WL_F64}


def _require_gpu(device) -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.WlError("waterlily_amd needs an AMD GPU (gfx950): torch.cuda.is_available() is False "
                           "and there is no CPU fallback")
    dev = torch.device(device)
    check(_lib.lib().wl_set_device(dev.index or 0))
    return dev


# --------------------------------------------------------------------------- device arrays

class Layout:
    """Strides of one grid level.  padded=True: row pitch is a multiple of 128 B and the allocation is
    offset so that element [1,j,k] (first interior cell of a row) is 128-B aligned; padded=False is the
    reference's dense column-major layout."""

    def __init__(self, Ng: Sequence[int], T, padded: bool = True, slab=None):
        """Ng: extents of the UNDECOMPOSED array (ghosts included).  slab (dist.Slab, D==3 only): this rank's
        z-slab -- the local array then has slab.n2l planes (interior share + 2-plane halos)."""
        self.slab = slab
        self.Ng_global = tuple(int(n) for n in Ng)
        self.Ng = tuple(int(n) for n in Ng) if slab is None else tuple(int(n) for n in Ng[:2]) + (slab.n2l,)
        self.D = len(self.Ng)
        self.T = np.dtype(T)
        n = self.Ng + (1,) * (3 - self.D)
        al = 128 // self.T.itemsize
        if padded:
            self.lead = al - 1
            sy = -(-n[0] // al) * al
        else:
            self.lead = 0
            sy = n[0]
        sz = sy * n[1]
        self.n3 = n
        self.s = (1, sy, sz)
        self.span = sz * n[2]
        self.sc = self.span
        self.align = al

    def grid(self) -> Grid:
        g = Grid()
        g.D = self.D
        g.n[:] = list(self.n3)
        g.s[:] = list(self.s)
        g.sc = self.sc
        if self.slab is not None:
            g.nzg, g.kz0, g.own_lo, g.own_hi = self.slab.nzg, self.slab.kz0, self.slab.own_lo, self.slab.own_hi
            g.zring = int(self.slab.ring)
        return g

    def alloc(self, ncomp: Sequence[int], device, fill: float = 0.0) -> torch.Tensor:
        """A zero (or `fill`) field with trailing component dims `ncomp` ((), (D,), (D,D))."""
        nc = int(np.prod(ncomp)) if len(ncomp) else 1
        buf = _arena_take(self.lead + nc * self.sc + self.align, _TORCH[self.T], device, fill)
        size = self.Ng + tuple(ncomp)
        stride = tuple(self.s[: self.D])
        cs = self.sc
        for _ in ncomp:
            stride += (cs,)
            cs *= ncomp[0]
        a = torch.as_strided(buf, size, stride, buf.storage_offset() + self.lead)
        a._wl_slab = self.slab
        return a


# Placement experiment (tools/placement.py): fields carved out of ONE device buffer at a chosen
# spacing instead of one torch allocation per field.  Off unless set_arena() was called.
_ARENA = {"buf": None, "pos": 0, "round": 1, "skew": 0}


def set_arena(nbytes: int, device, round_to: int = 1 << 21, skew: int = 0) -> None:
    """Carve every field allocated from now on out of one NEW `nbytes` buffer: field k starts at the next multiple of
    `round_to` after field k-1, plus `skew` bytes.  nbytes = 0 switches back to one torch allocation per field.
    (A fresh buffer per call: fields of earlier Simulations keep the previous buffer alive and are never overlapped.)"""
    if nbytes == 0:
        _ARENA.update(buf=None, pos=0)
        return
    _ARENA["buf"] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    _ARENA.update(pos=0, round=round_to, skew=skew)


def _arena_take(nelem: int, dtype, device, fill: float) -> torch.Tensor:
    a = _ARENA
    item = torch.empty((), dtype=dtype).element_size()
    if a["buf"] is None:
        return torch.full((nelem,), fill, dtype=dtype, device=device)
    start = -(-a["pos"] // a["round"]) * a["round"] + (a["skew"] if a["pos"] else 0)
    start = -(-start // 256) * 256
    if start + nelem * item > a["buf"].numel():          # does not fit any more: an allocation of its own
        return torch.full((nelem,), fill, dtype=dtype, device=device)
    a["pos"] = start + nelem * item
    out = a["buf"][start:start + nelem * item].view(dtype)
    out.fill_(fill)
    return out


def _grid_of(a: torch.Tensor, D: int) -> Grid:
    """Recover the wl_grid of a strided field view."""
    g = Grid()
    g.D = D
    n = tuple(a.shape[:D]) + (1,) * (3 - D)
    st = tuple(a.stride()[:D])
    g.n[:] = list(n)
    s = list(st) + [st[-1] * n[D - 1]] * (3 - D)
    g.s[:] = s
    g.sc = a.stride()[D] if a.ndim > D else s[D - 1] * n[D - 1]
    sl = getattr(a, "_wl_slab", None)
    if sl is not None:
        g.nzg, g.kz0, g.own_lo, g.own_hi = sl.nzg, sl.kz0, sl.own_lo, sl.own_hi
        g.zring = int(sl.ring)
    return g


def _T(a: torch.Tensor) -> np.dtype:
    return np.dtype(np.float32) if a.dtype == torch.float32 else np.dtype(np.float64)


def _ptr(a: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(a.data_ptr())


def like(a: torch.Tensor, fill: float = 0.0) -> torch.Tensor:
    """A new field with the SAME strides and row alignment as `a` (torch's .clone() would return a dense tensor,
    which must never be handed to the library together with `a`'s grid)."""
    item = a.element_size()
    al = 128 // item
    lead = (a.data_ptr() // item) % al
    need = 1 + sum((n - 1) * st for n, st in zip(a.shape, a.stride()))
    buf = torch.full((lead + need + al,), fill, dtype=a.dtype, device=a.device)
    out = torch.as_strided(buf, a.size(), a.stride(), lead)
    out._wl_slab = getattr(a, "_wl_slab", None)
    return out


def copy_of(a: torch.Tensor) -> torch.Tensor:
    b = like(a)
    b.copy_(a)
    return b


def _same_layout(a: torch.Tensor, b: torch.Tensor) -> None:
    if a.shape != b.shape or a.stride() != b.stride() or a.dtype != b.dtype:
        raise ValueError("fields must share shape, strides and dtype (use waterlily_amd.sim.like/copy_of, not .clone())")


def dot(a: torch.Tensor, b: torch.Tensor) -> float:
    """LinearAlgebra.dot over the whole arrays, ghost cells included (src/Poisson.jl:126-146)"""
    _same_layout(a, b)
    out = C.c_double()
    g = _grid_of(a, a.ndim)
    check(_lib.lib().wl_dot(_WLT[_T(a)], C.byref(g), _ptr(a), _ptr(b), C.byref(out)))
    return out.value


def divergence(z: torch.Tensor, u: torch.Tensor) -> None:
    """@inside z[I] = div(I,u)  (src/Flow.jl:139)"""
    if tuple(u.shape[:-1]) != tuple(z.shape) or u.stride()[:-1] != z.stride():
        raise ValueError("z and u must share the grid layout")
    g = _grid_of(z, z.ndim)
    check(_lib.lib().wl_div(_WLT[_T(z)], C.byref(g), _ptr(z), _ptr(u)))


def to_host(a: torch.Tensor) -> np.ndarray:
    """`Array(field)`: dense Fortran-ordered host copy."""
    return np.asfortranarray(a.detach().cpu().numpy())


def upload(a: torch.Tensor, h: np.ndarray) -> None:
    """`copyto!(field, host_array)`"""
    a.copy_(torch.from_numpy(np.ascontiguousarray(h)).to(a.dtype))


def permask(perdir: Sequence[int]) -> int:
    m = 0
    for j in perdir:
        m |= 1 << int(j)
    return m


# --------------------------------------------------------------------------- util.jl

def loc(i: int, shape: Sequence[int], kz0: int = 0) -> np.ndarray:
    """util.jl:160 for every index of an array of extents `shape` (i=-1: cell centre); kz0 = global index of
    local plane 0 when `shape` is a z-slab."""
    ax = [np.arange(n, dtype=np.float64) - 0.5 for n in shape]
    if kz0:
        ax[-1] = ax[-1] + kz0
    x = np.stack(np.meshgrid(*ax, indexing="ij"))
    if i >= 0:
        x[i] -= 0.5
    return x


def apply_vec(f: Callable, c: torch.Tensor) -> None:
    """util.jl:171 applyV! -- runs the user's closure on the host, then uploads."""
    D = c.ndim - 1
    h = np.zeros(tuple(c.shape), dtype=_T(c), order="F")
    sl = getattr(c, "_wl_slab", None)
    for i in range(D):
        x = loc(i, c.shape[:-1], sl.kz0 if sl is not None else 0)
        h[..., i] = np.broadcast_to(np.asarray(f(i, x), dtype=np.float64), c.shape[:-1])
    upload(c, h)


def inside(a) -> tuple:
    return tuple(slice(1, n - 1) for n in a.shape)


def L2(a: torch.Tensor) -> float:
    """util.jl:68 (the reference's GPU override: ext/WaterLilyAMDGPUExt.jl:24)"""
    out = C.c_double()
    g = _grid_of(a, a.ndim)
    check(_lib.lib().wl_L2_inside(_WLT[_T(a)], C.byref(g), _ptr(a), C.byref(out)))
    return out.value


def BC(a: torch.Tensor, A, saveexit: bool = False, perdir: Sequence[int] = ()) -> None:
    """util.jl:192-210"""
    g = _grid_of(a, a.ndim - 1)
    check(_lib.lib().wl_bc_vec(_WLT[_T(a)], C.byref(g), _ptr(a), d3(A), int(saveexit), permask(perdir)))


def perBC(a: torch.Tensor, perdir: Sequence[int]) -> None:
    """util.jl:227-231"""
    g = _grid_of(a, a.ndim)
    check(_lib.lib().wl_bc_per(_WLT[_T(a)], C.byref(g), _ptr(a), permask(perdir)))


def halo_exchange(a: torch.Tensor, depth: int = 2) -> None:
    """Fill the z-halo planes of a decomposed field from the neighbouring ranks (no-op when not decomposed)."""
    sl = getattr(a, "_wl_slab", None)
    if sl is None:
        return
    D = 3
    ncomp = int(np.prod(a.shape[D:])) if a.ndim > D else 1
    g = _grid_of(a, D)
    check(_lib.lib().wl_halo_exchange(_WLT[_T(a)], C.byref(g), _ptr(a), ncomp, depth))


def gather(a: torch.Tensor) -> np.ndarray:
    """Assemble the undecomposed host array from the owned planes of every rank (tests / output)."""
    sl = getattr(a, "_wl_slab", None)
    h = to_host(a)
    if sl is None:
        return h
    mine = np.ascontiguousarray(np.moveaxis(h, 2, 0)[sl.own_lo:sl.own_hi + 1])
    parts = [mine]
    if sl.size > 1:
        import torch.distributed as dist
        parts = [None] * sl.size
        dist.all_gather_object(parts, mine)
    full = np.concatenate(parts, axis=0)
    if sl.ring:   # nobody owns the two z ghost planes of a periodic ring: they are the wrapped interior planes
        full = np.concatenate([full[-1:], full, full[:1]], axis=0)
    return np.asfortranarray(np.moveaxis(full, 0, 2))


def exitBC(u: torch.Tensor, u0: torch.Tensor, U, dt: float) -> None:
    """util.jl:216-222"""
    _same_layout(u, u0)
    g = _grid_of(u, u.ndim - 1)
    check(_lib.lib().wl_exit_bc(_WLT[_T(u)], C.byref(g), _ptr(u), _ptr(u0), d3(U), float(dt)))


_exit_bc = exitBC


def BCTuple(U, dt: Sequence[float], D: int):
    """Flow.jl:79-80"""
    if callable(U):
        t = float(np.sum(np.asarray(dt, dtype=np.float64)))
        return tuple(float(U(i, t)) for i in range(D))
    return tuple(float(x) for x in U)


def _dUdt(U: Callable, i: int, t: float) -> float:
    # reference: ForwardDiff.derivative (Flow.jl:71-72); host-side scalar, central difference
    h = 1e-5 * max(1.0, abs(t))
    return (float(U(i, t + h)) - float(U(i, t - h))) / (2 * h)


def accel_tuple(g, U, dt: Sequence[float], D: int):
    """Flow.jl:68-73: g(i,t)+dU_i/dt at t=sum(dt); None when accelerate! is a no-op."""
    if g is None and not callable(U):
        return None
    t = float(np.sum(np.asarray(dt, dtype=np.float64)))
    return tuple((float(g(i, t)) if g is not None else 0.0) + (_dUdt(U, i, t) if callable(U) else 0.0)
                 for i in range(D))


# --------------------------------------------------------------------------- Flow.jl

class Flow:
    """src/Flow.jl:92-122.  Fields: u, u0, f, V, mu0 (Ng...,D); mu1 (Ng...,D,D); p, sigma (Ng...).
    u0 is scratch between steps, as in the reference (mom_step! overwrites it before reading it, Flow.jl:154): after a step of a
    3-D run without periodic directions / convective exit it holds the predictor's velocity, not the old one (wl_set_option(27))."""

    def __init__(self, N, U, *, dt=0.25, nu=0.0, g=None, ulam=None, perdir=(), exitBC=False, T=np.float64,
                 device="cuda:0", padded=True, slab=None):
        self.device = _require_gpu(device)
        D = len(N)
        self.D, self.T = D, np.dtype(T)
        Ng = tuple(int(n) + 2 for n in N)
        self.N = Ng                      # extents of the undecomposed arrays (the reference's size(p))
        self.slab = slab
        if slab is not None and D != 3:
            raise ValueError("z-slab decomposition needs D == 3")
        if slab is not None and slab.ring != (2 in tuple(perdir)):
            raise ValueError("Slab.ring must be set exactly when z (direction 2) is periodic")
        self.layout = Layout(Ng, T, padded, slab)
        self.U, self.g, self.nu = U, g, float(nu)
        self.perdir, self.exitBC = tuple(int(j) for j in perdir), bool(exitBC)
        self.dt = [float(self.T.type(dt))]
        al = lambda nc, fill=0.0: self.layout.alloc(nc, self.device, fill)
        self.u = al((D,))
        apply_vec(ulam if ulam is not None else (lambda i, x: 0.0), self.u)
        U0 = BCTuple(U, [0.0], D)
        BC(self.u, U0, exitBC, perdir)
        _exit_bc(self.u, self.u, U0, 0.0)
        halo_exchange(self.u, 2)
        self.u0 = al((D,))
        self.u0.copy_(self.u)
        self.f, self.p, self.sigma = al((D,)), al(()), al(())
        self.V, self.mu0, self.mu1 = al((D,)), al((D,), 1.0), al((D, D))
        BC(self.mu0, (0.0,) * D, False, perdir)
        halo_exchange(self.mu0, 2)
        desc = FlowDesc()
        desc.g = self.layout.grid()
        for k in ("u", "u0", "f", "p", "sigma", "V", "mu0", "mu1"):
            setattr(desc, k, getattr(self, k).data_ptr())
        desc.nu, desc.exitBC, desc.perdir_mask = self.nu, int(self.exitBC), permask(self.perdir)
        self._h = C.c_void_p()
        check(_lib.lib().wl_flow_create(C.byref(self._h), _WLT[self.T], C.byref(desc)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().wl_flow_destroy(self._h)
        except Exception:
            pass


def time(a: Flow) -> float:
    """Flow.jl:129"""
    return float(np.sum(np.asarray(a.dt[:-1], dtype=np.float64)))


def conv_diff(r: torch.Tensor, u: torch.Tensor, Phi=None, nu=0.1, perdir=()):
    """Flow.jl:36-51.  The gather kernels need no scratch; when Phi is given its top ghost cells receive what the reference's
    scatter form leaves there (include/wlhip.h: wl_conv_diff)."""
    _same_layout(r, u)
    g = _grid_of(u, u.ndim - 1)
    if Phi is not None and (tuple(Phi.shape) != tuple(u.shape[:-1]) or Phi.stride() != u.stride()[:-1]):
        raise ValueError("Phi and u must share the grid layout")
    check(_lib.lib().wl_conv_diff(_WLT[_T(u)], C.byref(g), _ptr(r), _ptr(u), None if Phi is None else _ptr(Phi), float(nu),
                                  permask(perdir)))


def accelerate(r: torch.Tensor, acc) -> None:
    g = _grid_of(r, r.ndim - 1)
    check(_lib.lib().wl_accelerate(_WLT[_T(r)], C.byref(g), _ptr(r), d3(acc)))


def BDIM(a: Flow) -> None:
    """Flow.jl:131-135"""
    g = a.layout.grid()
    check(_lib.lib().wl_bdim(_WLT[a.T], C.byref(g), _ptr(a.u), _ptr(a.u0), _ptr(a.f), _ptr(a.V), _ptr(a.mu0),
                             _ptr(a.mu1), a.dt[-1]))


def scale_u(a: Flow, scale: float) -> None:
    """Flow.jl:170"""
    g = a.layout.grid()
    check(_lib.lib().wl_scale_u(_WLT[a.T], C.byref(g), _ptr(a.u), float(scale)))


def CFL(a: Flow) -> float:
    """Flow.jl:172-175"""
    out = C.c_double()
    g = a.layout.grid()
    check(_lib.lib().wl_cfl(_WLT[a.T], C.byref(g), _ptr(a.sigma), _ptr(a.u), a.nu, C.byref(out)))
    return out.value


# --------------------------------------------------------------------------- Poisson.jl / MultiLevelPoisson.jl

class _Level:
    """One `Poisson` (src/Poisson.jl:21-30): L, D, iD, x, eps, r, z."""

    def __init__(self, lay: Layout, x, L, z, device):
        self.layout, self.x, self.L, self.z = lay, x, L, z
        self.D, self.iD, self.eps, self.r = (lay.alloc((), device) for _ in range(4))
        self.shape = lay.Ng_global

    def desc(self) -> LevelDesc:
        d = LevelDesc()
        d.g = self.layout.grid()
        for k in ("L", "D", "iD", "x", "eps", "r", "z"):
            setattr(d, k, getattr(self, k).data_ptr())
        return d


def _layout_of(x: torch.Tensor) -> Layout:
    D = x.ndim
    sl = getattr(x, "_wl_slab", None)
    shape = tuple(x.shape) if sl is None else tuple(x.shape[:2]) + (sl.nzg,)
    lay = Layout(shape, _T(x), padded=False, slab=sl)
    st = tuple(x.stride())
    n = lay.n3
    lay.s = (1, st[1], st[2] if D == 3 else st[1] * n[1])
    lay.span = lay.s[2] * n[2]
    lay.sc = lay.span
    # keep the alignment rule of the fields we were given: first interior element of a row on a 128-B boundary
    if (x.data_ptr() + x.element_size()) % 128 == 0:
        lay.lead = lay.align - 1
    return lay


class _PoissonBase:
    def _create(self, levels, perdir):
        self.levels = levels
        arr = (LevelDesc * len(levels))(*[l.desc() for l in levels])
        self._h = C.c_void_p()
        check(_lib.lib().wl_mg_create(C.byref(self._h), _WLT[self.T], len(levels), arr, permask(perdir)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().wl_mg_destroy(self._h)
        except Exception:
            pass


class Poisson(_PoissonBase):
    """src/Poisson.jl:21-38 (single level)"""

    def __init__(self, x: torch.Tensor, L: torch.Tensor, z: torch.Tensor, perdir=()):
        assert x.shape == z.shape and tuple(L.shape) == tuple(x.shape) + (x.ndim,)
        self.T = _T(x)
        self.x, self.L, self.z, self.perdir = x, L, z, tuple(perdir)
        self.n: list[int] = []
        lay = _layout_of(x)
        lay.sc = L.stride(-1)
        lv = _Level(lay, x, L, z, x.device)
        self._create([lv], perdir)
        self.D, self.iD, self.eps, self.r = lv.D, lv.iD, lv.eps, lv.r


class MultiLevelPoisson(_PoissonBase):
    """src/MultiLevelPoisson.jl:44-60"""

    def __init__(self, x: torch.Tensor, L: torch.Tensor, z: torch.Tensor, maxlevels=10, perdir=(), padded=True,
                 replicate_cells=1 << 21):
        """replicate_cells (multi-GPU only): levels with at most this many interior cells are replicated on every
        rank instead of being z-slabs (their kernels are launch-latency bound, and a replicated level needs no
        halo exchange and no all-reduce per dot product)."""
        assert x.shape == z.shape and tuple(L.shape) == tuple(x.shape) + (x.ndim,)
        self.T = _T(x)
        D = x.ndim
        self.x, self.L, self.z, self.perdir = x, L, z, tuple(perdir)
        self.n: list[int] = []
        lay = _layout_of(x)
        lay.sc = L.stride(-1)
        levels = [_Level(lay, x, L, z, x.device)]
        # restrictML, MultiLevelPoisson.jl:18-25,53-55; which levels stay z-slabs and which are replicated: dist.plan_levels
        from .dist import plan_levels
        for Na, slab in plan_levels(lay.Ng_global, lay.slab, maxlevels, replicate_cells)[1:]:
            la = Layout(Na, self.T, padded, slab)
            levels.append(_Level(la, la.alloc((), x.device), la.alloc((D,), x.device), la.alloc((), x.device), x.device))
        if len(levels) <= 2:
            raise AssertionError("MultiLevelPoisson requires size=a2ⁿ, where n>2")
        self._create(levels, perdir)


def update(p, flow: Optional["Flow"] = None) -> None:
    """Poisson.jl:46 / MultiLevelPoisson.jl:62-68.  flow: the Flow whose mu0 is p.L -- after a native measure! of it only
    the rows that measure! rewrote are revisited on the finest level (same values; wl_mg_update_changed)."""
    if flow is not None and hasattr(p, "_h") and hasattr(flow, "_h"):
        check(_lib.lib().wl_mg_update_changed(p._h, flow._h))
    else:
        check(_lib.lib().wl_mg_update(p._h))


def mult(p, x: torch.Tensor) -> torch.Tensor:
    """Poisson.jl:62-68"""
    assert x.shape == p.z.shape and x.stride() == p.z.stride()
    check(_lib.lib().wl_mg_mult(p._h, 0, _ptr(x)))
    return p.z


def residual(p, level=0):
    check(_lib.lib().wl_mg_residual(p._h, level))


def increment(p, level=0):
    check(_lib.lib().wl_mg_increment(p._h, level))


def Jacobi(p, it=1, level=0):
    check(_lib.lib().wl_mg_jacobi(p._h, level, it))


def uniform_rows(p, level=0):
    """(rows whose face coefficients are one number, owned interior rows) of a level -- see wl_mg_uniform_rows."""
    a, b = C.c_longlong(), C.c_longlong()
    check(_lib.lib().wl_mg_uniform_rows(p._h, int(level), C.byref(a), C.byref(b)))
    return a.value, b.value


def set_option(key: int, value: int):
    """Tuning / A-B switches of the library (include/wlhip.h, wl_set_option)."""
    check(_lib.lib().wl_set_option(int(key), int(value)))


def get_option(key: int) -> int:
    v = C.c_int()
    check(_lib.lib().wl_get_option(int(key), C.byref(v)))
    return v.value


def pcg(p, it=6, level=0) -> int:
    n = C.c_int()
    check(_lib.lib().wl_mg_pcg(p._h, level, it, C.byref(n)))
    return n.value


def L2p(p, level=0) -> float:
    """Poisson.jl:146"""
    out = C.c_double()
    check(_lib.lib().wl_mg_L2(p._h, level, C.byref(out)))
    return out.value


def Linf(p, level=0) -> float:
    """Poisson.jl:147"""
    out = C.c_double()
    check(_lib.lib().wl_mg_Linf(p._h, level, C.byref(out)))
    return out.value


def solver_log(p, on=True) -> None:
    """Switch the reference's pressure-solver log (`@log`, util.jl:4-24; Poisson.jl:164,167; MultiLevelPoisson.jl:90,94)
    on or off for this hierarchy; read_solver_log returns the rows (n, L∞, L₂) recorded since the last read."""
    check(_lib.lib().wl_mg_log(p._h, int(bool(on))))


def read_solver_log(p, cap=4096) -> np.ndarray:
    rows = (C.c_double * (3 * cap))()
    n = C.c_int()
    check(_lib.lib().wl_mg_log_read(p._h, rows, cap, C.byref(n)))
    return np.array(rows[:3 * min(n.value, cap)], dtype=np.float64).reshape(-1, 3)


def format_solver_log(rows, prefix="") -> str:
    """the text WaterLily.logger writes: header "p/c, iter, r∞, r₂" (util.jl:23), then `prefix` ("p" / "c", Flow.jl:158,165)
    in front of the n = 0 row of each solve"""
    out = []
    for n, rinf, r2 in rows:
        out.append(f"{prefix if n == 0 else ''}, {int(n)}, {rinf}, {r2}\n")
    return "".join(out)


def comm_counts() -> dict:
    """collectives this rank issued since the last wl_prof_reset (include/wlhip.h: wl_prof_comm)"""
    v = (C.c_int64 * 6)()
    check(_lib.lib().wl_prof_comm(v))
    return dict(zip(("allreduce", "exchanges", "sendrecv_pairs", "allgather", "halo_bytes", "allgather_bytes"), (int(x) for x in v)))


def Vcycle(ml: MultiLevelPoisson, l=0):
    check(_lib.lib().wl_mg_vcycle(ml._h, l))


def solver(p, tol=1e-4, itmx=None):
    """Poisson.jl:162-172 / MultiLevelPoisson.jl:87-99"""
    if itmx is None:
        itmx = 32 if isinstance(p, MultiLevelPoisson) else 1000
    n = C.c_int()
    check(_lib.lib().wl_mg_solve(p._h, float(tol), int(itmx), C.byref(n)))
    p.n.append(n.value)


def _two(a, b, D_a, D_b):
    return _grid_of(a, D_a), _grid_of(b, D_b)


def restrict(a: torch.Tensor, b: torch.Tensor):
    ga, gb = _two(a, b, a.ndim, b.ndim)
    check(_lib.lib().wl_restrict(_WLT[_T(a)], C.byref(ga), _ptr(a), C.byref(gb), _ptr(b)))


def prolongate(a: torch.Tensor, b: torch.Tensor):
    ga, gb = _two(a, b, a.ndim, b.ndim)
    check(_lib.lib().wl_prolongate(_WLT[_T(a)], C.byref(ga), _ptr(a), C.byref(gb), _ptr(b)))


def restrictL(a: torch.Tensor, b: torch.Tensor, perdir=()):
    ga, gb = _two(a, b, a.ndim - 1, b.ndim - 1)
    check(_lib.lib().wl_restrictL(_WLT[_T(a)], C.byref(ga), _ptr(a), C.byref(gb), _ptr(b), permask(perdir)))


def project(a: Flow, b: MultiLevelPoisson, w=1.0) -> int:
    """Flow.jl:137-145"""
    n = C.c_int()
    check(_lib.lib().wl_project(a._h, b._h, a.dt[-1], float(w), C.byref(n)))
    b.n.append(n.value)
    return n.value


def mom_step(a: Flow, b: MultiLevelPoisson) -> None:
    """Flow.jl:153-169"""
    U = BCTuple(a.U, a.dt, a.D)
    gp = accel_tuple(a.g, a.U, a.dt[:-1], a.D)
    gc = accel_tuple(a.g, a.U, a.dt, a.D)
    n2 = (C.c_int * 2)()
    dtn = C.c_double()
    check(_lib.lib().wl_mom_step(a._h, b._h, a.dt[-1], d3(U), None if gp is None else d3(gp),
                                 None if gc is None else d3(gc), C.byref(dtn), n2))
    b.n.extend([int(n2[0]), int(n2[1])])
    a.dt.append(float(dtn.value))


# --------------------------------------------------------------------------- WaterLily.jl

class Simulation:
    """src/WaterLily.jl:59-79"""

    def __init__(self, dims, u_BC, L, *, dt=0.25, nu=0.0, g=None, U=None, eps=1, perdir=(), ulam=None,
                 exitBC=False, body=None, T=np.float32, device="cuda:0", padded=True, slab="auto",
                 replicate_cells=1 << 21, geometry="device"):
        """geometry: where the body's sdf/map closures are evaluated by measure! -- "device": on the GPU, written
        straight into mu0/mu1/V (no host round trip; closures must be device-agnostic torch code); "host": on the
        CPU then uploaded (bit-identical to what the CPU oracle is fed in the parity tests)."""
        assert not (callable(u_BC) and callable(ulam)), "`u_BC` and `uλ` cannot be both specified as Function"
        assert not (U is None and callable(u_BC)), "`U` must be specified if `u_BC` is a Function"
        if ulam is None:
            ulam = (lambda i, x: u_BC(i, 0.0)) if callable(u_BC) else (lambda i, x: u_BC[i])
        self.U = float(np.sqrt(sum(float(v) ** 2 for v in u_BC))) if U is None else U
        self.L, self.eps = L, eps
        self.body = body if body is not None else B.NoBody()
        self.geometry = geometry
        if slab == "auto":   # one z-slab per rank once a communicator exists (waterlily_amd.dist.init_*)
            from . import dist as _dist
            r, n = _dist.rank_size()
            slab = _dist.Slab(r, n, int(dims[2]), ring=(2 in tuple(perdir))) if (n > 1 and len(dims) == 3) else None
        self.slab = slab
        self.flow = Flow(dims, u_BC, ulam=ulam, dt=dt, nu=nu, g=g, T=T, perdir=perdir, exitBC=exitBC,
                         device=device, padded=padded, slab=slab)
        self._band = None
        measure_flow(self.flow, self.body, t=0.0, eps=eps, geometry=geometry)
        self.pois = MultiLevelPoisson(self.flow.p, self.flow.mu0, self.flow.sigma, perdir=perdir, padded=padded,
                                      replicate_cells=replicate_cells)


def measure_flow(a: Flow, body, t=0.0, eps=1, geometry="device") -> None:
    """Body.jl:31-53: the user's sdf/map closures are evaluated with torch (autograd = ForwardDiff) either on the
    GPU, straight into the coefficient fields, or on the host followed by an upload; the two BC! calls, the halo
    exchange and the row flags run in the library."""
    if isinstance(body, B.NoBody):
        flow_update(a)
        return
    dims = tuple(n - 2 for n in a.N)
    a._band_cells = None
    if geometry == "device" and B.is_native(body):
        # closed-form family + affine map: the whole measure! (fill loop, both BC! calls, halos, row flags) runs as
        # hand-written kernels (csrc/wl_measure.h); only rows holding body cells now or before are rewritten
        L = _lib.lib()
        desc = body.native_desc(t, a.D)
        nband = C.c_int64()
        check(L.wl_measure_rows(a._h, desc, float(eps), C.byref(nband)))
        cand = torch.empty(max(1, nband.value), dtype=torch.int64, device=a.device)
        check(L.wl_measure_fill(a._h, desc, float(eps), C.c_void_p(cand.data_ptr())))
        a._band_cells = (float(t), cand[:nband.value])
        return
    if geometry == "device":
        cells = B.measure_fields_into(body, dims, a.mu0, a.mu1, a.V, a.sigma, t=t, eps=eps, slab=a.slab)
        a._band_cells = (float(t), cells)          # reused by pressure_force at the same body time
    else:
        mu0, mu1, V, d = B.measure_fields(body, dims, t=t, eps=eps, T=a.T, slab=a.slab)
        upload(a.mu0, mu0)
        upload(a.mu1, mu1)
        upload(a.V, V)
        a.sigma[inside(a.sigma)] = torch.from_numpy(np.ascontiguousarray(d[inside(d)])).to(a.sigma.device)
    BC(a.mu0, (0.0,) * a.D, False, a.perdir)
    BC(a.V, (0.0,) * a.D, a.exitBC, a.perdir)
    halo_exchange(a.mu0, 2)   # (mu1 needs no exchange: the host evaluated the halo planes from the sdf directly)
    halo_exchange(a.V, 2)
    flow_update(a)


def flow_update(a: Flow) -> None:
    """Tell the library that mu0/mu1/V changed (rebuilds BDIM!'s body-free row flags); measure! does it."""
    check(_lib.lib().wl_flow_update(a._h))


def sim_time(sim: Simulation) -> float:
    """WaterLily.jl:89"""
    return time(sim.flow) * sim.U / sim.L


def measure(sim: Simulation, t=None) -> None:
    """WaterLily.jl:116-119"""
    t = float(np.sum(np.asarray(sim.flow.dt, dtype=np.float64))) if t is None else t
    measure_flow(sim.flow, sim.body, t=t, eps=sim.eps, geometry=sim.geometry)
    sim._band = None
    update(sim.pois, sim.flow)


def sim_step(sim: Simulation, t_end=None, *, remeasure=True, max_steps=None, verbose=False) -> None:
    """WaterLily.jl:98-109"""
    if t_end is None:
        sim._moving = bool(remeasure)          # remeasure=False: the body is declared static (see _ensure_band)
        if remeasure:
            measure(sim)
        mom_step(sim.flow, sim.pois)
        return
    steps0 = len(sim.flow.dt)
    while sim_time(sim) < t_end and (max_steps is None or len(sim.flow.dt) - steps0 < max_steps):
        sim_step(sim, remeasure=remeasure)
        if verbose:
            print(f"tU/L={sim_time(sim):.4f}, Δt={sim.flow.dt[-1]:.3f}")


def pressure_force_band(p: torch.Tensor, idx: torch.Tensor, nds: torch.Tensor) -> np.ndarray:
    """Metrics.jl:94-100 with the body term handed over as a compact band (see include/wlhip.h)."""
    D = p.ndim
    out = (C.c_double * 3)()
    g = _grid_of(p, D)
    check(_lib.lib().wl_pforce(_WLT[_T(p)], C.byref(g), _ptr(p), C.c_void_p(idx.data_ptr()),
                               C.c_void_p(nds.data_ptr()), idx.numel(), out))
    return np.array(out[:D])


def band_to_device(p: torch.Tensor, idx: np.ndarray, nds: np.ndarray):
    """Translate dense column-major cell indices (body.nds_band) into element offsets of the strided
    field `p` and upload both arrays."""
    D = p.ndim
    sub = np.unravel_index(idx, tuple(p.shape), order="F")
    off = sum(s.astype(np.int64) * int(st) for s, st in zip(sub, p.stride()))
    return (torch.from_numpy(np.ascontiguousarray(off)).to(p.device),
            torch.from_numpy(np.ascontiguousarray(nds, dtype=np.float64)).to(p.device))


# --------------------------------------------------------------------------- Metrics.jl field metrics

_METRIC = {"ke": 0, "curl": 1, "omega_mag": 2, "omega_theta": 3, "lambda2": 4}


def metric(out: torch.Tensor, kind: str, u: torch.Tensor, i: int = 0, par=None, par2=None) -> torch.Tensor:
    """`@inside out[I] = ke(I,u,U) | curl(i,I,u) | ω_mag(I,u) | ω_θ(I,z,center,u) | λ₂(I,u)` (src/Metrics.jl:14-77).
    kind in {"ke","curl","omega_mag","omega_theta","lambda2"}; i = 0-based curl component; par = U (ke) or z (ω_θ),
    par2 = center (ω_θ)."""
    if tuple(u.shape[:-1]) != tuple(out.shape) or u.stride()[:-1] != out.stride():
        raise ValueError("out and u must share the grid layout")
    g = _grid_of(out, out.ndim)
    check(_lib.lib().wl_metric(_WLT[_T(out)], C.byref(g), _METRIC[kind], _ptr(out), _ptr(u), int(i),
                               None if par is None else d3(par), None if par2 is None else d3(par2)))
    return out


def _band_of(sim: Simulation):
    """(idx, nds) device tensors of the |d|<=1 band at the current flow time (cached per time)"""
    _ensure_band(sim)
    return sim._band[1], sim._band[2]


def viscous_force(sim: Simulation) -> np.ndarray:
    """Metrics.jl:108-113"""
    idx, nds = _band_of(sim)
    u = sim.flow.u
    out = (C.c_double * 3)()
    g = _grid_of(u, u.ndim - 1)
    check(_lib.lib().wl_vforce(_WLT[_T(u)], C.byref(g), _ptr(u), C.c_void_p(idx.data_ptr()), C.c_void_p(nds.data_ptr()),
                               idx.numel(), sim.flow.nu, out))
    return np.array(out[:sim.flow.D])


def total_force(sim: Simulation) -> np.ndarray:
    """Metrics.jl:120"""
    return pressure_force(sim) + viscous_force(sim)


def pressure_moment(x0, sim: Simulation) -> np.ndarray:
    """Metrics.jl:128-134"""
    idx, nds = _band_of(sim)
    p = sim.flow.p
    out = (C.c_double * 3)()
    g = _grid_of(p, p.ndim)
    check(_lib.lib().wl_pmoment(_WLT[_T(p)], C.byref(g), _ptr(p), C.c_void_p(idx.data_ptr()), C.c_void_p(nds.data_ptr()),
                                idx.numel(), d3(x0), out))
    return np.array(out[:sim.flow.D])


def pressure_force(sim: Simulation) -> np.ndarray:
    """Metrics.jl:94-95"""
    _ensure_band(sim)
    return pressure_force_band(sim.flow.p, sim._band[1], sim._band[2])


def _ensure_band(sim: Simulation) -> None:
    """The |d| <= 1 band (cell offsets + n*kern vectors, Metrics.jl:84-87) on the device.  The reference evaluates
    nds(body, x, t) over the whole grid at every call; here it is rebuilt only when it can have changed: when the flow
    time moved AND the body is being re-measured (sim_step(remeasure=True), a moving body).  With remeasure=False the
    caller declares the body static -- the solver keeps using the coefficients of the last measure! -- and the band of
    that measure! stays valid.  A rebuild examines only the cells of the last measure!'s band |d| < 2+eps (a body moves
    less than a cell per step), on the device, without a host round trip."""
    t = time(sim.flow)
    if sim._band is not None and (sim._band[0] == t or not getattr(sim, "_moving", True)):
        return
    dims = tuple(n - 2 for n in sim.flow.N)
    bc = getattr(sim.flow, "_band_cells", None)
    if bc is not None and sim.geometry == "device" and B.is_native(sim.body):
        cand = bc[1]
        if sim.slab is not None:                       # owned interior planes only (the force is all-reduced)
            kk = cand // int(sim.flow.N[0] * sim.flow.N[1])
            sl = sim.slab
            cand = cand[(kk >= sl.own_lo) & (kk <= sl.own_hi) & (kk + sl.kz0 >= 1) & (kk + sl.kz0 <= sim.flow.N[2] - 2)]
        desc = sim.body.native_desc(t, sim.flow.D)
        nds = torch.empty((cand.numel(), sim.flow.D), dtype=torch.float64, device=cand.device)
        g = _grid_of(sim.flow.p, sim.flow.D)
        check(_lib.lib().wl_body_nds(C.byref(g), desc, C.c_void_p(cand.data_ptr()), cand.numel(), C.c_void_p(nds.data_ptr())))
        keep = (nds != 0).any(1)
        sim._band = (t,) + band_to_device_t(sim.flow.p, cand[keep], nds[keep])
        return
    if bc is not None and sim.geometry == "device":
        idx, nds = B.nds_band_from_candidates(sim.body, dims, bc[1], t=t, slab=sim.slab)
        sim._band = (t,) + band_to_device_t(sim.flow.p, idx, nds)
        return
    idx, nds = B.nds_band(sim.body, dims, t=t, slab=sim.slab,
                          device=sim.flow.device if sim.geometry == "device" else "cpu")
    sim._band = (t,) + band_to_device(sim.flow.p, idx, nds)


def band_to_device_t(p: torch.Tensor, idx: torch.Tensor, nds: torch.Tensor):
    """band_to_device for index / vector tensors that already live on p's device"""
    off = torch.zeros_like(idx)
    rem = idx.clone()
    dense = np.cumprod((1,) + tuple(p.shape[:-1]))
    for ddim in range(p.ndim - 1, -1, -1):
        off += (rem // int(dense[ddim])) * int(p.stride()[ddim])
        rem = rem % int(dense[ddim])
    return off.contiguous(), nds.to(torch.float64).contiguous()
