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
import os
from typing import Callable, Optional, Sequence, Tuple

import numpy as np
import torch

__all__ = ["AutoBody", "NoBody", "Sphere", "Torus", "AffineMap", "translation", "rotation2d", "measure", "sdf", "kern",
           "kern0", "kern1", "mu0", "mu1", "norm2", "measure_fields", "measure_fields_into", "nds_band"]


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
        # (Julia's min(x,y) = ifelse(isless(y,x), y, x): at a tie the value AND the derivative of x -- torch.minimum would split the gradient)
        sd = lambda x, t: (lambda sa, sb: torch.where(sb < sa, sb, sa))(a.sdf(x, t), b.sdf(x, t))
        out = AutoBody(sd, mp, compose=False)
        out.identity_map = a.identity_map and b.identity_map
        return out

    __or__ = __add__

    def __and__(a, b: "AutoBody") -> "AutoBody":
        mp = lambda x, t: torch.where(a.sdf(x, t) > b.sdf(x, t), a.map(x, t), b.map(x, t))
        sd = lambda x, t: (lambda sa, sb: torch.where(sa < sb, sb, sa))(a.sdf(x, t), b.sdf(x, t))
        out = AutoBody(sd, mp, compose=False)
        out.identity_map = a.identity_map and b.identity_map
        return out

    def __neg__(a) -> "AutoBody":
        out = AutoBody(lambda x, t: -a.sdf(x, t), a.map, compose=False)
        out.identity_map = a.identity_map
        return out

    def __sub__(a, b: "AutoBody") -> "AutoBody":
        return a & (-b)


# --- parametric bodies: closed-form sdf family + affine map, measured by the HIP kernels of csrc/wl_measure.h --------
class AffineMap:
    """map(x,t) = A(t) x + b(t) with its exact time derivative (what ForwardDiff.derivative gives the reference,
    AutoBody.jl:129): `coeffs(t)` returns numpy (A (D,D), b (D,), dA/dt, db/dt)."""

    def __init__(self, coeffs: Callable):
        self.coeffs = coeffs

    def __call__(self, x, t):   # the torch closure of the same map (generic / host path)
        A, b, _, _ = self.coeffs_torch(x, t)
        return A @ x + b[:, None]

    def coeffs_torch(self, x, t):
        raise NotImplementedError


class _PolyTranslation(AffineMap):
    """xi = x - (s0 + v t + a t^2): `move` (v=(1,0)) and `accel` (a=(2,0)) of maintests.jl:373-374"""

    def __init__(self, D, v=0.0, a=0.0, s0=0.0):
        self.D = D
        self.v, self.a, self.s0 = (np.broadcast_to(np.asarray(q, dtype=np.float64), (D,)).copy() for q in (v, a, s0))

    def coeffs(self, t):
        I, Z = np.eye(self.D), np.zeros((self.D, self.D))
        return I, -(self.s0 + self.v * t + self.a * t * t), Z, -(self.v + 2 * self.a * t)

    def __call__(self, x, t):
        cols = [torch.as_tensor(float(self.s0[i]), dtype=x.dtype, device=x.device) + float(self.v[i]) * t + float(self.a[i]) * t * t
                for i in range(self.D)]
        return x - torch.stack(cols)[:, None]


def translation(D, v=0.0, a=0.0, s0=0.0) -> AffineMap:
    return _PolyTranslation(D, v, a, s0)


class _Rotation2D(AffineMap):
    """xi = R(theta) (x - c), R = [c s; -s c], theta = w t + th0 (maintests.jl:376-379)"""

    def __init__(self, center, w, th0=0.0):
        self.c, self.w, self.th0 = float(center), float(w), float(th0)

    def coeffs(self, t):
        s, c = math.sin(self.w * t + self.th0), math.cos(self.w * t + self.th0)
        R, dR = np.array([[c, s], [-s, c]]), np.array([[-s, c], [-c, -s]]) * self.w
        cc = np.full(2, self.c)
        return R, -R @ cc, dR, -dR @ cc

    def __call__(self, x, t):
        s, c = torch.sin(self.w * t + self.th0), torch.cos(self.w * t + self.th0)
        e = x - self.c
        return torch.stack([c * e[0] + s * e[1], -s * e[0] + c * e[1]])


def rotation2d(center, w, th0=0.0) -> AffineMap:
    return _Rotation2D(center, w, th0)


class ParametricBody(AutoBody):
    """An AutoBody whose sdf is one of the library's closed-form families and whose map (if any) is affine: besides the
    torch closures (generic path: host geometry, other back ends) it can describe itself to the HIP `measure!` kernel."""
    family = -1

    def __init__(self, sdf_closure, params, map: Optional[AffineMap] = None):
        super().__init__(sdf_closure, map)
        self.params, self.amap = [float(v) for v in params], map

    def _fill_desc(self, d, t: float, D: int) -> None:
        d.family, d.identity_map = self.family, int(self.amap is None)
        for q, v in enumerate(self.params):
            d.p[q] = v
        A, b, dA, db = (np.eye(D), np.zeros(D), np.zeros((D, D)), np.zeros(D)) if self.amap is None else self.amap.coeffs(float(t))
        Ai = np.linalg.inv(A)
        for name, M in (("A", A), ("dA", dA), ("Ainv", Ai)):
            full = np.zeros((3, 3))
            full[:D, :D] = M
            getattr(d, name)[:] = list(full.ravel())
        for name, vec in (("b", b), ("db", db)):
            full = np.zeros(3)
            full[:D] = vec
            getattr(d, name)[:] = list(full)

    def native_desc(self, t: float, D: int):
        """the wl_body_desc array (one element) the HIP measure! kernels take"""
        from ._lib import BodyDesc
        arr = (BodyDesc * 1)()
        self._fill_desc(arr[0], t, D)
        arr[0].count = 1
        return arr

    # set operations on parametric bodies stay parametric: a `Bodies` composite the HIP kernels measure natively
    def __add__(a, b):
        return Bodies([a], []) + b if isinstance(b, (ParametricBody, Bodies)) else AutoBody.__add__(a, b)

    __or__ = __add__

    def __and__(a, b):
        return Bodies([a], []) & b if isinstance(b, (ParametricBody, Bodies)) else AutoBody.__and__(a, b)

    def __sub__(a, b):
        return Bodies([a], []) - b if isinstance(b, (ParametricBody, Bodies)) else AutoBody.__sub__(a, b)


class Bodies(AutoBody):
    """`Bodies(bodies, ops)` (src/AutoBody.jl:40-110) of PARAMETRIC leaves: combined left to right, ops[i-1] in "+", "-", "&"
    ("∪" = "+", "∩" = "&") between the composite of bodies[:i] and bodies[i].  Its torch closures (generic path) are the fold
    of the AutoBody operators (AutoBody.jl:22-34); on the device the whole composite is ONE wl_body_desc array."""
    _OPS = {"+": 0, "∪": 0, "|": 0, "-": 1, "&": 2, "∩": 2}

    def __init__(self, bodies, ops=None):
        bodies = list(bodies)
        ops = ["+"] * (len(bodies) - 1) if ops is None else list(ops)
        if len(bodies) != len(ops) + 1:
            raise ValueError("length(bodies) != length(ops)+1")
        if any(o not in self._OPS for o in ops):
            raise ValueError("Operations array `ops` not supported. Use only `ops ∈ [+,-,∩,∪]`")
        if not all(isinstance(b, ParametricBody) for b in bodies):
            raise TypeError("Bodies takes parametric leaves (Sphere, Cylinder, Torus, Plate); closures combine through AutoBody's operators")
        self.bodies, self.ops = bodies, ops
        acc = bodies[0]
        for b, o in zip(bodies[1:], ops):
            f = {0: AutoBody.__add__, 1: AutoBody.__sub__, 2: AutoBody.__and__}[self._OPS[o]]
            acc = f(acc, b)
        self.sdf, self.map, self.identity_map = acc.sdf, acc.map, acc.identity_map

    def _with(self, other, op):
        if isinstance(other, ParametricBody):
            return Bodies(self.bodies + [other], self.ops + [op])
        if isinstance(other, Bodies):
            # the leaves of `other` may only be appended to the left fold where that keeps the grouping: a single leaf, or a
            # union of unions (associative).  a - (b + c), a & (b + c), a + (b - c), a + (b & c) ... are NOT left folds of
            # the leaves: like the reference's AutoBody operators (AutoBody.jl:22-34) they nest, as closures
            union_of_unions = op == "+" and all(self._OPS[o] == 0 for o in other.ops)
            if len(other.bodies) > 1 and not union_of_unions:
                return {"+": AutoBody.__add__, "-": AutoBody.__sub__, "&": AutoBody.__and__}[op](self, other)
            return Bodies(self.bodies + other.bodies, self.ops + [op] + other.ops)
        return {"+": AutoBody.__add__, "-": AutoBody.__sub__, "&": AutoBody.__and__}[op](self, other)

    def __add__(self, other):
        return self._with(other, "+")

    __or__ = __add__

    def __and__(self, other):
        return self._with(other, "&")

    def __sub__(self, other):
        return self._with(other, "-")

    def native_desc(self, t: float, D: int):
        from ._lib import WL_BODY_MAXLEAF, BodyDesc
        n = len(self.bodies)
        if n > WL_BODY_MAXLEAF:
            raise ValueError(f"a native composite holds at most {WL_BODY_MAXLEAF} leaves")
        arr = (BodyDesc * n)()
        for l, b in enumerate(self.bodies):
            b._fill_desc(arr[l], t, D)
            arr[l].op = self._OPS[self.ops[l - 1]] if l else 0
        arr[0].count = n
        return arr


def is_native(body) -> bool:
    """can the HIP measure! kernels take this body? (a parametric leaf or a composite of at most WL_BODY_MAXLEAF of them)"""
    from ._lib import WL_BODY_MAXLEAF
    return isinstance(body, ParametricBody) or (isinstance(body, Bodies) and len(body.bodies) <= WL_BODY_MAXLEAF)


class Sphere(ParametricBody):
    """sqrt(sum(abs2, x .- center)) - radius: circle (2-D) / sphere (3-D) of README.md:41-44,118-120.  `center`: a
    scalar (every axis) or one value per axis."""
    family = 0

    def __init__(self, center, radius, D: int, map: Optional[AffineMap] = None):
        c = np.broadcast_to(np.asarray(center, dtype=np.float64), (D,)).copy()
        cl = [float(v) for v in c]

        def sdf_closure(x, t):
            return torch.sqrt(sum((x[i] - cl[i]) ** 2 for i in range(D))) - radius
        super().__init__(sdf_closure, list(np.concatenate([c, np.zeros(3 - D)])) + [radius], map)


class Cylinder(ParametricBody):
    """sqrt(sum over the axes in `axes` of (x - center)^2) - radius: a circle extruded along the remaining axes (default: the
    last one) -- the 3-D cylinder of the reference's examples/ThreeD_cylinder*.jl, `norm2(x[1:2] .- center) - radius`."""
    family = 3

    def __init__(self, center, radius, D: int = 3, axes=None, map: Optional[AffineMap] = None):
        axes = tuple(range(D - 1)) if axes is None else tuple(int(a) for a in axes)
        c = np.broadcast_to(np.asarray(center, dtype=np.float64), (D,)).copy()
        cl = [float(v) for v in c]
        mask = [1.0 if a in axes else 0.0 for a in range(3)]

        def sdf_closure(x, t):
            return torch.sqrt(sum((x[i] - cl[i]) ** 2 for i in axes)) - radius
        super().__init__(sdf_closure, list(np.concatenate([c, np.zeros(3 - D)])) + [radius] + mask, map)


class Torus(ParametricBody):
    """norm((x1-c1, norm((x2-c2, x3-c3)) - R)) - r: the "donut" """
    family = 1

    def __init__(self, center, R, r, map: Optional[AffineMap] = None):
        c = [float(v) for v in np.broadcast_to(np.asarray(center, dtype=np.float64), (3,))]

        def sdf_closure(x, t):
            q = torch.sqrt((x[1] - c[1]) ** 2 + (x[2] - c[2]) ** 2) - R
            return torch.sqrt((x[0] - c[0]) ** 2 + q ** 2) - r
        super().__init__(sdf_closure, c + [R, r], map)


class Plate(ParametricBody):
    """norm(x - (clamp(x1,-a,a), 0[, 0])) - thk: the reference's test plate (test/maintests.jl:375), a stadium (2-D) or
    capsule (3-D) of half-span `a` and half-thickness `thk` about the first axis of the mapped coordinates."""
    family = 2

    def __init__(self, a, thk, D: int, map: Optional[AffineMap] = None):
        a, thk = float(a), float(thk)

        def sdf_closure(x, t):
            e0 = x[0] - torch.clamp(x[0], -a, a)
            return torch.sqrt(e0 ** 2 + sum(x[i] ** 2 for i in range(1, D))) - thk
        super().__init__(sdf_closure, [a, thk], map)


def _as_points(x) -> Tuple[torch.Tensor, bool]:
    x = torch.as_tensor(np.asarray(x, dtype=np.float64)) if not isinstance(x, torch.Tensor) else x.to(torch.float64)
    single = x.ndim == 1
    return (x[:, None] if single else x), single


def sdf(body: AutoBody, x, t=0.0, keep_dtype=False) -> torch.Tensor:
    """AutoBody.jl:38.  keep_dtype: evaluate in the dtype of the tensor `x` (the reference evaluates sdf(loc(0,I,T)) in
    the field type T, Body.jl:34) instead of Float64."""
    if keep_dtype and isinstance(x, torch.Tensor) and x.is_floating_point():
        xp, single = (x[:, None] if x.ndim == 1 else x), x.ndim == 1
    else:
        xp, single = _as_points(x)
    # (t in the points' dtype: the reference passes t::T (Body.jl:31), and a Float64 t would promote a map closure's
    #  output -- and the whole evaluation -- to Float64)
    d = body.sdf(xp, torch.as_tensor(float(t), dtype=xp.dtype, device=xp.device))
    d = torch.broadcast_to(d, (xp.shape[1],))
    return d[0] if single else d


def measure(body: AutoBody, x, t=0.0, fastd2: float = math.inf):
    """AutoBody.jl:110-131: returns (d, n, V); n, V are zero where d^2 > fastd2.

    d is corrected to a pseudo-sdf (d/|grad|), n is the unit normal, V = -J^-1 * dmap/dt.
    Runs on the device of `x` (CPU or GPU); torch autograd stands in for ForwardDiff."""
    xp, single = _as_points(x)
    D, M = xp.shape
    dev = xp.device
    tt = torch.as_tensor(float(t), dtype=torch.float64, device=dev)
    xr = xp.detach().clone().requires_grad_(True)
    d = torch.broadcast_to(body.sdf(xr, tt), (M,))
    n = torch.zeros(D, M, dtype=torch.float64, device=dev)
    V = torch.zeros(D, M, dtype=torch.float64, device=dev)
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
            J = torch.zeros(M, D, D, dtype=torch.float64, device=dev)
            for a in range(D):
                if mo.requires_grad:
                    (ga,) = torch.autograd.grad(mo[a].sum(), xq, retain_graph=True, allow_unused=True)
                    if ga is not None:
                        J[:, a, :] = ga.T
            _, dot = torch.func.jvp(lambda s: body.map(xp, s), (tt,), (torch.ones_like(tt),))
            dot = torch.broadcast_to(dot, (D, M))
            eye = torch.eye(D, dtype=torch.float64, device=dev)[None]
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


# torch twins of the moments (any device)
def _kern_t(d):
    return 0.5 + 0.5 * torch.cos(math.pi * d)


def _mu0_t(d, eps):
    d = torch.clamp(d / eps, -1, 1)
    return 0.5 + 0.5 * d + 0.5 * torch.sin(math.pi * d) / math.pi


def _mu1_t(d, eps):
    d = torch.clamp(d / eps, -1, 1)
    return eps * (0.25 * (1 - d * d) - 0.5 * (d * torch.sin(math.pi * d) + (1 + torch.cos(math.pi * d)) / math.pi) / math.pi)


# --- field-level measure (Body.jl:31-50) --------------------------------------------------------------

_TT = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}


def _chunk_points(Ng, lo, hi, device, dtype=torch.float64):
    """cell centres loc(0,I) (util.jl:160) of the INTERIOR cells with global last index in [lo,hi):
    returns (points (D, M) of `dtype`, shape of the chunk).  (Half-integers: exact in Float32 up to 2^23.)"""
    ax = [torch.arange(1, n - 1, dtype=dtype, device=device) - 0.5 for n in Ng[:-1]]
    ax.append(torch.arange(lo, hi, dtype=dtype, device=device) - 0.5)
    g = torch.meshgrid(*ax, indexing="ij")
    shp = tuple(g[0].shape)
    return torch.stack([q.reshape(-1) for q in g]), shp


def _fill_whole(a: "torch.Tensor", value) -> None:
    """a.fill_(value).  A padded field (sim.Layout.alloc) is a strided view of ONE flat buffer of its own; filling the
    view runs torch's generic strided kernel (0.9 ms per 512^3 component), filling the flat buffer -- row padding
    included, nothing reads it -- is a plain memset-speed stream (15 components: 14 ms -> 2 ms per measure!)."""
    base = a._base
    if base is not None and base.dim() == 1 and base.is_contiguous() and a.numel() <= base.numel() <= 2 * a.numel():
        base.fill_(value)
    else:
        a.fill_(value)


def measure_fields_into(body, dims: Sequence[int], mu0, mu1, V, dsdf, t: float = 0.0, eps: float = 1.0,
                        chunk_cells: Optional[int] = None, slab=None):
    """Body.jl:31-50 before the two BC! calls, written INTO the given torch tensors (any device, any strides):
    mu0 (Nl...,D), mu1 (Nl...,D,D), V (Nl...,D), dsdf (Nl...) where Nl are the local extents (= Ng without a slab).
    The user's sdf/map closures run on the tensors' device: on the GPU this is `measure!` without a host round trip.
    Returns the LOCAL column-major linear indices (int64 tensor) of the band cells d^2 < (2+eps)^2, which
    `nds_band(candidates=...)` can reuse so that pressure_force does not have to scan the grid again."""
    D = len(dims)
    Ng = tuple(int(n) + 2 for n in dims)      # extents of the undecomposed array
    dev, tdt = mu0.device, mu0.dtype
    if chunk_cells is None:                    # big chunks on the GPU: few launches, few host syncs
        chunk_cells = int(os.environ.get("WL_MEASURE_CHUNK", 1 << 28)) if dev.type == "cuda" else (1 << 22)   # 512^3 in one piece
    lstrides = [1]
    for n in mu0.shape[:D - 1]:
        lstrides.append(lstrides[-1] * int(n))
    cand = []
    kz0 = slab.kz0 if slab is not None else 0
    nl = mu0.shape[D - 1]
    _fill_whole(mu0, 1)
    _fill_whole(mu1, 0)
    _fill_whole(V, 0)
    if body is None or isinstance(body, NoBody):
        return torch.zeros(0, dtype=torch.int64, device=dev)
    d2 = float((2 + eps) ** 2)
    plane = int(np.prod(Ng[:-1]))
    step = max(1, chunk_cells // plane)
    g_lo, g_hi = max(1, kz0), min(Ng[-1] - 1, kz0 + nl)   # global interior planes present locally
    inner = tuple(slice(1, n - 1) for n in Ng[:-1])
    for lo in range(g_lo, g_hi, step):
        hi = min(g_hi, lo + step)
        # the distance of every cell centre, evaluated in the field type T like the reference (Body.jl:34: sdf(loc(0,I,T)),
        # stored into sigma::T) -- for Float32 fields half the bytes of a Float64 evaluation over the whole grid.  It
        # only decides band membership (|d| < 2+eps, where mu0 = 1 and mu1 = V = 0 anyway) and inside/outside far from
        # the surface; the band cells themselves are measured in Float64 below.
        pts, shp = _chunk_points(Ng, lo, hi, dev, tdt)
        dc = sdf(body, pts, t, keep_dtype=True).to(tdt)
        ksl = slice(lo - kz0, hi - kz0)
        dsdf[inner + (ksl,)] = dc.reshape(shp)
        band = (dc * dc) < torch.as_tensor(d2, dtype=tdt, device=dev)   # Body.jl:35, compared in T
        inside_body = (~band) & (dc < 0)
        iidx = torch.nonzero(inside_body)[:, 0]             # cells inside the body, away from the surface: mu0 = 0
        if iidx.numel():                                     # (scatter on the few such cells, not a masked pass over the field)
            isub = list(torch.unravel_index(iidx, shp))
            ifull = tuple(q + 1 for q in isub[:-1]) + (isub[-1] + (lo - kz0),)
            for i in range(D):
                mu0[ifull + (i,)] = 0
        if bool(band.any()):
            bidx = torch.nonzero(band)[:, 0]
            xb = pts[:, bidx].to(torch.float64)
            sub = list(torch.unravel_index(bidx, shp))
            full = tuple(s + 1 for s in sub[:-1]) + (sub[-1] + (lo - kz0),)
            cand.append(sum(f.to(torch.int64) * int(s) for f, s in zip(full, lstrides)))
            for i in range(D):
                xf = xb.clone()
                xf[i] -= 0.5                                  # face location loc(i,I) (util.jl:160)
                di, ni, Vi = measure(body, xf, t, fastd2=d2)
                V[full + (i,)] = Vi[i].to(tdt)
                mu0[full + (i,)] = _mu0_t(di, eps).to(tdt)
                k1 = _mu1_t(di, eps)
                for j in range(D):
                    mu1[full + (i, j)] = (k1 * ni[j]).to(tdt)
    return torch.cat(cand) if cand else torch.zeros(0, dtype=torch.int64, device=dev)


def _fortran_empty(shape, tdt):
    st, acc = [], 1
    for n in shape:
        st.append(acc)
        acc *= n
    return torch.empty_strided(tuple(shape), tuple(st), dtype=tdt)


def measure_fields(body, dims: Sequence[int], t: float = 0.0, eps: float = 1.0, T=np.float32,
                   chunk_cells: int = 1 << 22, slab=None):
    """Host form of measure_fields_into: returns Fortran-ordered numpy arrays (mu0, mu1, V, d)."""
    D = len(dims)
    Ng = tuple(int(n) + 2 for n in dims)
    Nl = Ng if slab is None else Ng[:-1] + (slab.n2l,)
    tdt = _TT[np.dtype(T)]
    m0, m1 = _fortran_empty(Nl + (D,), tdt), _fortran_empty(Nl + (D, D), tdt)
    Vv, dd = _fortran_empty(Nl + (D,), tdt), _fortran_empty(Nl, tdt)
    dd.zero_()
    measure_fields_into(body, dims, m0, m1, Vv, dd, t=t, eps=eps, chunk_cells=chunk_cells, slab=slab)
    return m0.numpy(), m1.numpy(), Vv.numpy(), dd.numpy()


def nds_band_from_candidates(body, dims: Sequence[int], candidates: "torch.Tensor", t: float = 0.0, slab=None):
    """nds_band restricted to `candidates` (LOCAL column-major linear indices, a superset of the |d| <= 1 band, on the
    device where the closures shall run), everything kept on that device: returns (idx int64, nds (n, D) float64)
    torch tensors sorted by idx -- no host round trip (measure!'s band cells -> pressure_force on the GPU)."""
    D = len(dims)
    Ng = tuple(int(n) + 2 for n in dims)
    strides = np.cumprod((1,) + Ng[:-1])
    dev = candidates.device
    kz0c = slab.kz0 if slab is not None else 0
    cand = candidates
    if slab is not None:                       # owned interior planes only
        kk = cand // int(strides[D - 1])
        cand = cand[(kk >= slab.own_lo) & (kk <= slab.own_hi) & (kk + kz0c >= 1) & (kk + kz0c <= Ng[-1] - 2)]
    if cand.numel() == 0:
        return torch.zeros(0, dtype=torch.int64, device=dev), torch.zeros((0, D), dtype=torch.float64, device=dev)
    rem, coords = cand.clone(), []
    for ddim in range(D - 1, -1, -1):
        coords.insert(0, rem // int(strides[ddim]))
        rem = rem % int(strides[ddim])
    pts = torch.stack([c.to(torch.float64) - 0.5 for c in coords])
    pts[D - 1] += kz0c
    d, n, _ = measure(body, pts, t, fastd2=1.0)
    v = (n * _kern_t(torch.clamp(d, -1, 1))[None]).T
    keep = (v != 0).any(1)
    idx, order = torch.sort(cand[keep], stable=True)
    return idx, v[keep][order].contiguous()


def nds_band(body, dims: Sequence[int], t: float = 0.0, chunk_cells: Optional[int] = None, slab=None, device="cpu",
             candidates=None):
    """Metrics.jl:84-87 evaluated over inside(p): returns (idx, nds) where idx are the column-major
    linear indices (ghost-inclusive LOCAL extents) of cells with a non-zero n*kern(clamp(d,-1,1)) and nds is
    the (nband, D) Float64 array of those vectors (positions and normals in Float64, Metrics.jl:96).
    Returned as numpy arrays; `device` only selects where the closures are evaluated.
    candidates (optional): local linear indices of a superset of the band (what measure_fields_into returned for the
    same t): only those cells are examined instead of scanning the whole grid."""
    D = len(dims)
    Ng = tuple(int(n) + 2 for n in dims)
    strides = np.cumprod((1,) + Ng[:-1])
    if body is None or isinstance(body, NoBody):
        return np.zeros(0, dtype=np.int64), np.zeros((0, D))
    if chunk_cells is None:
        chunk_cells = (1 << 25) if torch.device(device).type == "cuda" else (1 << 22)
    if candidates is not None:
        idx_t, nds_t = nds_band_from_candidates(body, dims, candidates.to(device), t=t, slab=slab)
        return idx_t.cpu().numpy(), np.ascontiguousarray(nds_t.cpu().numpy())
    plane = int(np.prod(Ng[:-1]))
    step = max(1, chunk_cells // plane)
    idxs, vals = [], []
    # z-slab: only the interior planes this rank OWNS (the force is all-reduced); indices are local
    kz0 = slab.kz0 if slab is not None else 0
    g_lo = 1 if slab is None else max(1, slab.kz0 + slab.own_lo)
    g_hi = Ng[-1] - 1 if slab is None else min(Ng[-1] - 1, slab.kz0 + slab.own_hi + 1)
    for lo in range(g_lo, g_hi, step):
        hi = min(g_hi, lo + step)
        pts, shp = _chunk_points(Ng, lo, hi, device)
        dc = sdf(body, pts, t)
        near = torch.nonzero(dc * dc <= 1.0 + 1e-9)[:, 0]      # generous pre-filter; exact test in measure
        if near.numel() == 0:
            continue
        d, n, _ = measure(body, pts[:, near], t, fastd2=1.0)
        v = (n * _kern_t(torch.clamp(d, -1, 1))[None]).T        # (m, D)
        keep = (v != 0).any(1)
        sub = torch.unravel_index(near[keep], shp)
        full = [s + 1 for s in sub[:-1]] + [sub[-1] + (lo - kz0)]
        lin = sum(f.to(torch.int64) * int(s) for f, s in zip(full, strides))
        idxs.append(lin.cpu().numpy())
        vals.append(v[keep].cpu().numpy())
    if not idxs:
        return np.zeros(0, dtype=np.int64), np.zeros((0, D))
    idx = np.concatenate(idxs)
    nds = np.concatenate(vals)
    order = np.argsort(idx, kind="stable")
    return idx[order], np.ascontiguousarray(nds[order])
