"""Host-side immersed-body geometry: AutoBody, measure, BDIM kernel moments.

Mirrors /root/reference/src/AutoBody.jl:13-131 and src/Body.jl:31-61.  In the reference `measure!`
evaluates *user closures* (sdf, map) with ForwardDiff inside a generated kernel; closures cannot cross a
C ABI, so -- as SURVEY.md section 8 row a3 prescribes -- geometry stays on the host: closures are evaluated
in Float64 with torch (CPU) and autograd plays the role of ForwardDiff.  The resulting coefficient fields
(mu0, mu1, V) and the compact pressure-force band (n*kern(d)) are uploaded once per `measure!`.

Closure convention: ``sdf(x, t)`` and ``map(x, t)`` receive ``x`` as a float64 tensor of shape
``(D, M)`` (M query points; ``x[0]`` is the reference's ``x[1]``) and ``t`` as a 0-dim float64 tensor;
``sdf`` returns shape ``(M,)``, ``map`` returns ``(D, M)``.  Only torch operations may be used.
"""
from __future__ import annotations

import math
from typing import Callable, Optional, Sequence, Tuple

import numpy as np
import torch

__all__ = ["AutoBody", "NoBody", "measure", "sdf", "kern", "kern0", "kern1", "mu0", "mu1", "norm2",
           "measure_fields", "nds_band"]


def norm2(x: torch.Tensor) -> torch.Tensor:
    """sqrt(sum(abs2, x)) over the coordinate axis."""
    return torch.sqrt((x * x).sum(0))


class NoBody:
    """Body.jl:75-76"""


class AutoBody:
    """AutoBody.jl:13-20: implicit geometry from ``sdf`` and an optional coordinate ``map``."""

    def __init__(self, sdf: Callable, map: Optional[Callable] = None, compose: bool = True):
        self.identity_map = map is None
        self.map = (lambda x, t: x) if map is None else map
        m = self.map
        self.sdf = (lambda x, t: sdf(m(x, t), t)) if (compose and map is not None) else sdf

    # AutoBody.jl:22-34 set operations
    def __add__(a, b: "AutoBody") -> "AutoBody":
        mp = lambda x, t: torch.where(a.sdf(x, t) < b.sdf(x, t), a.map(x, t), b.map(x, t))
        sd = lambda x, t: torch.minimum(a.sdf(x, t), b.sdf(x, t))
        out = AutoBody(sd, mp, compose=False)
        out.identity_map = a.identity_map and b.identity_map
        return out

    __or__ = __add__

    def __and__(a, b: "AutoBody") -> "AutoBody":
        mp = lambda x, t: torch.where(a.sdf(x, t) > b.sdf(x, t), a.map(x, t), b.map(x, t))
        sd = lambda x, t: torch.maximum(a.sdf(x, t), b.sdf(x, t))
        out = AutoBody(sd, mp, compose=False)
        out.identity_map = a.identity_map and b.identity_map
        return out

    def __neg__(a) -> "AutoBody":
        out = AutoBody(lambda x, t: -a.sdf(x, t), a.map, compose=False)
        out.identity_map = a.identity_map
        return out

    def __sub__(a, b: "AutoBody") -> "AutoBody":
        return a & (-b)


def _as_points(x) -> Tuple[torch.Tensor, bool]:
    x = torch.as_tensor(np.asarray(x, dtype=np.float64)) if not isinstance(x, torch.Tensor) else x.to(torch.float64)
    single = x.ndim == 1
    return (x[:, None] if single else x), single


def sdf(body: AutoBody, x, t=0.0) -> torch.Tensor:
    """AutoBody.jl:38"""
    xp, single = _as_points(x)
    d = body.sdf(xp, torch.as_tensor(float(t), dtype=torch.float64))
    d = torch.broadcast_to(d, (xp.shape[1],))
    return d[0] if single else d


def measure(body: AutoBody, x, t=0.0, fastd2: float = math.inf):
    """AutoBody.jl:110-131: returns (d, n, V); n, V are zero where d^2 > fastd2.

    d is corrected to a pseudo-sdf (d/|grad|), n is the unit normal, V = -J^-1 * dmap/dt."""
    xp, single = _as_points(x)
    D, M = xp.shape
    tt = torch.as_tensor(float(t), dtype=torch.float64)
    xr = xp.detach().clone().requires_grad_(True)
    d = torch.broadcast_to(body.sdf(xr, tt), (M,))
    n = torch.zeros(D, M, dtype=torch.float64)
    V = torch.zeros(D, M, dtype=torch.float64)
    dv = d.detach().clone()
    near = dv * dv <= fastd2
    if bool(near.any()):
        if d.requires_grad:
            (g,) = torch.autograd.grad(d.sum(), xr, allow_unused=True)
            g = torch.zeros_like(xr) if g is None else g
        else:
            g = torch.zeros_like(xr)
        ok = near & ~torch.isnan(g).any(0)
        m = torch.sqrt((g * g).sum(0))
        msafe = torch.where(ok, m, torch.ones_like(m))
        dv = torch.where(ok, dv / msafe, dv)
        n = torch.where(ok[None], g / msafe[None], n)
        if not body.identity_map:
            # J[a,b] = d map_a / d x_b ; dot = d map / d t  (forward-mode along t)
            xq = xp.detach().clone().requires_grad_(True)
            mo = body.map(xq, tt)
            J = torch.zeros(M, D, D, dtype=torch.float64)
            for a in range(D):
                if mo.requires_grad:
                    (ga,) = torch.autograd.grad(mo[a].sum(), xq, retain_graph=True, allow_unused=True)
                    if ga is not None:
                        J[:, a, :] = ga.T
            _, dot = torch.func.jvp(lambda s: body.map(xp, s), (tt,), (torch.ones_like(tt),))
            dot = torch.broadcast_to(dot, (D, M))
            eye = torch.eye(D, dtype=torch.float64)[None]
            Js = torch.where(ok[:, None, None], J, eye)
            Vs = -torch.linalg.solve(Js, dot.T[..., None])[..., 0]
            V = torch.where(ok[None], Vs.T, V)
    if single:
        return dv[0], n[:, 0], V[:, 0]
    return dv, n, V


# --- Body.jl:55-61 convolution kernel and its moments (Float64) -----------------------------------

def kern(d):
    return 0.5 + 0.5 * np.cos(np.pi * d)


def kern0(d):
    return 0.5 + 0.5 * d + 0.5 * np.sin(np.pi * d) / np.pi


def kern1(d):
    return 0.25 * (1 - d ** 2) - 0.5 * (d * np.sin(np.pi * d) + (1 + np.cos(np.pi * d)) / np.pi) / np.pi


def mu0(d, eps):
    return kern0(np.clip(np.asarray(d, dtype=np.float64) / eps, -1, 1))


def mu1(d, eps):
    return eps * kern1(np.clip(np.asarray(d, dtype=np.float64) / eps, -1, 1))


# --- field-level measure (Body.jl:31-50) --------------------------------------------------------------

def _centres(Ng: Sequence[int], lo: int, hi: int) -> np.ndarray:
    """loc(0,I) (util.jl:160) for all I with last index in [lo,hi): array (D, n0, .., hi-lo)."""
    D = len(Ng)
    ax = [np.arange(n, dtype=np.float64) - 0.5 for n in Ng[:-1]] + [np.arange(lo, hi, dtype=np.float64) - 0.5]
    return np.stack(np.meshgrid(*ax, indexing="ij"))


def measure_fields(body, dims: Sequence[int], t: float = 0.0, eps: float = 1.0, T=np.float32,
                   chunk_cells: int = 1 << 22, slab=None):
    """Body.jl:31-50 before the two BC! calls: returns host arrays (mu0, mu1, V, d), Fortran order,
    shaped (Ng...,D), (Ng...,D,D), (Ng...,D), (Ng...).  `d` holds sdf at the cell centres (the
    reference stores it in flow.sigma).  Cells outside `inside(p)` keep mu0=1, mu1=V=0."""
    D = len(dims)
    Ng = tuple(int(n) + 2 for n in dims)      # extents of the undecomposed array
    T = np.dtype(T)
    # z-slab (waterlily_amd.dist.Slab): local arrays hold the global planes kz0 .. kz0+n2l-1, halos included --
    # they are evaluated directly from the sdf, so no exchange is needed for the coefficient fields
    kz0 = slab.kz0 if slab is not None else 0
    Nl = Ng if slab is None else Ng[:-1] + (slab.n2l,)
    m0 = np.ones(Nl + (D,), dtype=T, order="F")
    m1 = np.zeros(Nl + (D, D), dtype=T, order="F")
    Vv = np.zeros(Nl + (D,), dtype=T, order="F")
    dd = np.zeros(Nl, dtype=T, order="F")
    if body is None or isinstance(body, NoBody):
        return m0, m1, Vv, dd
    d2 = T.type((2 + eps) ** 2)
    plane = int(np.prod(Ng[:-1]))
    step = max(1, chunk_cells // plane)
    g_lo, g_hi = max(1, kz0), min(Ng[-1] - 1, kz0 + Nl[-1])   # global interior planes present locally
    for lo in range(g_lo, g_hi, step):
        hi = min(g_hi, lo + step)
        xc = _centres(Ng, lo, hi)
        inner = tuple(slice(1, n - 1) for n in Ng[:-1]) + (slice(None),)
        xc = xc[(slice(None),) + inner]                     # interior cells of this chunk
        shp = xc.shape[1:]
        pts = torch.from_numpy(np.ascontiguousarray(xc.reshape(D, -1)))
        dc = sdf(body, pts, t).numpy().astype(T)             # stored into sigma::T (Body.jl:34)
        sel = inner[:-1] + (slice(lo - kz0, hi - kz0),)
        dd[sel] = dc.reshape(shp)
        band = (dc * dc) < d2                                 # Body.jl:35, compared in T
        inside_body = (~band) & (dc < 0)
        if inside_body.any():
            for i in range(D):
                v = m0[sel + (i,)]
                v[inside_body.reshape(shp)] = 0
                m0[sel + (i,)] = v
        if band.any():
            bidx = np.nonzero(band)[0]
            xb = np.ascontiguousarray(xc.reshape(D, -1)[:, bidx])
            sub = np.unravel_index(bidx, shp)
            full = tuple(s + 1 for s in sub[:-1]) + (sub[-1] + lo - kz0,)
            for i in range(D):
                xf = xb.copy()
                xf[i] -= 0.5                                  # face location loc(i,I) (util.jl:160)
                di, ni, Vi = measure(body, torch.from_numpy(xf), t, fastd2=float(d2))
                di, ni, Vi = di.numpy(), ni.numpy(), Vi.numpy()
                Vv[full + (i,)] = Vi[i].astype(T)
                m0[full + (i,)] = mu0(di, eps).astype(T)
                k1 = mu1(di, eps)
                for j in range(D):
                    m1[full + (i, j)] = (k1 * ni[j]).astype(T)
    return m0, m1, Vv, dd


def nds_band(body, dims: Sequence[int], t: float = 0.0, chunk_cells: int = 1 << 22, slab=None):
    """Metrics.jl:84-87 evaluated over inside(p): returns (idx, nds) where idx are the column-major
    linear indices (ghost-inclusive extents) of cells with a non-zero n*kern(clamp(d,-1,1)) and nds is
    the (nband, D) Float64 array of those vectors (positions and normals in Float64, Metrics.jl:96)."""
    D = len(dims)
    Ng = tuple(int(n) + 2 for n in dims)
    strides = np.cumprod((1,) + Ng[:-1])
    if body is None or isinstance(body, NoBody):
        return np.zeros(0, dtype=np.int64), np.zeros((0, D))
    plane = int(np.prod(Ng[:-1]))
    step = max(1, chunk_cells // plane)
    idxs, vals = [], []
    # z-slab: only the interior planes this rank OWNS (the force is all-reduced); indices are local
    kz0 = slab.kz0 if slab is not None else 0
    g_lo = 1 if slab is None else max(1, slab.kz0 + slab.own_lo)
    g_hi = Ng[-1] - 1 if slab is None else min(Ng[-1] - 1, slab.kz0 + slab.own_hi + 1)
    for lo in range(g_lo, g_hi, step):
        hi = min(g_hi, lo + step)
        xc = _centres(Ng, lo, hi)
        inner = tuple(slice(1, n - 1) for n in Ng[:-1]) + (slice(None),)
        xc = xc[(slice(None),) + inner]
        shp = xc.shape[1:]
        pts = np.ascontiguousarray(xc.reshape(D, -1))
        dc = sdf(body, torch.from_numpy(pts), t).numpy()
        near = np.nonzero(dc * dc <= 1.0 + 1e-9)[0]           # generous pre-filter; exact test below
        if near.size == 0:
            continue
        d, n, _ = measure(body, torch.from_numpy(np.ascontiguousarray(pts[:, near])), t, fastd2=1.0)
        d, n = d.numpy(), n.numpy()
        v = (n * kern(np.clip(d, -1, 1))[None]).T           # (m, D)
        keep = np.any(v != 0, axis=1)
        sub = np.unravel_index(near[keep], shp)
        full = [s + 1 for s in sub[:-1]] + [sub[-1] + lo - kz0]
        lin = sum(f.astype(np.int64) * int(s) for f, s in zip(full, strides))
        idxs.append(lin)
        vals.append(v[keep])
    if not idxs:
        return np.zeros(0, dtype=np.int64), np.zeros((0, D))
    idx = np.concatenate(idxs)
    nds = np.concatenate(vals)
    order = np.argsort(idx, kind="stable")
    return idx[order], np.ascontiguousarray(nds[order])
