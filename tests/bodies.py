"""Paired bodies for the parity tests: the SAME geometry once as the product sees it (waterlily_amd.body.AutoBody: user
closures in torch, differentiated by autograd -- or a parametric body the HIP `measure!` kernel understands) and once as
the oracle sees it (oracle.geometry.Body: closed-form sdf / map derivatives in numpy).  A parity test hands `.product`
to waterlily_amd and `.oracle` to oracle.wl_oracle, so no coefficient field the oracle uses was computed by product code.
The reference bodies these restate: README.md:41-44,118-120 (circle / sphere), SURVEY.md 8d C5 (torus),
test/maintests.jl:371-384 (move, accel, plate + rotate / bend)."""
from collections import namedtuple

import torch

from oracle import geometry as G
from waterlily_amd import body as B
from waterlily_amd.body import AutoBody, norm2

# product: closures (torch + autograd); oracle: closed forms (numpy); native: the same body as a waterlily_amd parametric
# body, measured by the HIP kernels of csrc/wl_measure.h (None where the family / map has no native form)
Twin = namedtuple("Twin", "product oracle native", defaults=(None,))


def _shift(f):
    """map(x,t) = x - (f(t), 0[, 0])"""
    def m(x, t):
        return x - torch.stack([f(t)] + [torch.zeros_like(t)] * (x.shape[0] - 1))[:, None]
    return m


def sphere(center, radius):
    """circle (2-D) / sphere (3-D): sqrt(sum(abs2, x .- center)) - radius"""
    def sdf(x, t):   # (center: one number for every axis, or one per axis)
        c = center if not hasattr(center, "__len__") else torch.as_tensor(center, dtype=x.dtype, device=x.device)[:, None]
        return norm2(x - c) - radius
    return Twin(AutoBody(sdf), G.Body(G.Sphere(center, radius)), lambda D: B.Sphere(center, radius, D))


def cylinder(center, radius, D=3):
    """circle extruded along the last axis: the reference's 3-D cylinder examples, norm2(x[1:2] .- center) - radius"""
    ax = tuple(range(D - 1))
    return Twin(AutoBody(lambda x, t: torch.sqrt(sum((x[i] - center) ** 2 for i in ax)) - radius),
                G.Body(G.Cylinder(center, radius, ax)), lambda Dn: B.Cylinder(center, radius, Dn))


def drilled_sphere(c=13.0, R=8.0, hole=3.0, cut=(13.0, 13.0, 17.5), v=0.0):
    """(sphere - cylinder) ∩ sphere2, the first sphere translating along x with speed v: AutoBody's -, ∩ (AutoBody.jl:22-34)
    as closures, the reference's `Bodies` (AutoBody.jl:40-110) on the oracle side and as a native composite"""
    mv = _shift(lambda t: v * t) if v else None
    vv = (v, 0.0, 0.0)
    prod = (AutoBody(lambda x, t: norm2(x - c) - R, mv) - AutoBody(lambda x, t: torch.sqrt((x[0] - c) ** 2 + (x[1] - c) ** 2) - hole)) \
        & AutoBody(lambda x, t: torch.sqrt(sum((x[i] - cut[i]) ** 2 for i in range(3))) - R)
    orc = G.Bodies([G.Body(G.Sphere(c, R), G.Translate(v=vv) if v else None), G.Body(G.Cylinder(c, hole, (0, 1))),
                    G.Body(G.Sphere(cut, R))], ["-", "&"])
    nat = lambda D: (B.Sphere(c, R, 3, map=B.translation(3, v=vv) if v else None) - B.Cylinder(c, hole, 3)) & B.Sphere(cut, R, 3)
    return Twin(prod, orc, nat)


def two_circles(v=1.5):
    """union of a fixed and a translating circle (2-D)"""
    prod = AutoBody(lambda x, t: norm2(x - 11.0) - 4.0) + AutoBody(lambda x, t: norm2(x - 14.3) - 3.1, _shift(lambda t: v * t))
    orc = G.Bodies([G.Body(G.Sphere(11.0, 4.0)), G.Body(G.Sphere(14.3, 3.1), G.Translate(v=(v, 0.0)))], ["+"])
    return Twin(prod, orc, lambda D: B.Sphere(11.0, 4.0, 2) + B.Sphere(14.3, 3.1, 2, map=B.translation(2, v=(v, 0.0))))


def torus(c, R, r):
    def sdf(x, t):
        q = torch.sqrt((x[1] - c) ** 2 + (x[2] - c) ** 2) - R
        return torch.sqrt((x[0] - c) ** 2 + q ** 2) - r
    return Twin(AutoBody(sdf), G.Body(G.Torus(c, R, r)), lambda D: B.Torus(c, R, r))


def moving_circle(center, radius, v=0.0, a=0.0, D=2):
    """circle / sphere translating along x: map(x,t) = x - (v t + a t^2, 0[, 0])   (maintests.jl:373-374: move v=1; accel a=2)"""
    vv, aa = (v,) + (0.0,) * (D - 1), (a,) + (0.0,) * (D - 1)
    return Twin(AutoBody(lambda x, t: norm2(x - center) - radius, _shift(lambda t: v * t + a * t ** 2)),
                G.Body(G.Sphere(center, radius), G.Translate(v=vv, a=aa)),
                lambda Dn: B.Sphere(center, radius, Dn, map=B.translation(Dn, v=vv, a=aa)))


def rotating_circle(center, radius, pivot, w, th0=0.0):
    """2-D circle whose coordinates rotate about `pivot`: map(x,t) = R(w t + th0) (x - pivot), sdf centred at `center`"""
    def rotate(x, t):
        s, c = torch.sin(w * t + th0), torch.cos(w * t + th0)
        e = x - pivot
        return torch.stack([c * e[0] + s * e[1], -s * e[0] + c * e[1]])
    return Twin(AutoBody(lambda x, t: norm2(x - center) - radius, rotate),
                G.Body(G.Sphere(center, radius), G.Rotate2D(pivot, w, th0)),
                lambda D: B.Sphere(center, radius, 2, map=B.rotation2d(pivot, w, th0)))


def _plate(radius):
    return lambda x, t: norm2(x - torch.stack([torch.clamp(x[0], -radius + 2, radius - 2), torch.zeros_like(x[0])])) - 2


def rotating_plate(radius):
    """maintests.jl:375-379"""
    def rotate(x, t):
        s, c = torch.sin(t / radius + 1), torch.cos(t / radius + 1)
        e = x - 2 * radius
        return torch.stack([c * e[0] + s * e[1], -s * e[0] + c * e[1]])
    return Twin(AutoBody(_plate(radius), rotate), G.Body(G.Plate(radius - 2, 2.0), G.Rotate2D(2 * radius, 1 / radius, 1.0)),
                lambda D: B.Plate(radius - 2, 2.0, 2, map=B.rotation2d(2 * radius, 1 / radius, 1.0)))


def bending_plate(radius):
    """maintests.jl:375,380-383"""
    def bend(xy, t):
        x, y = xy[0] - 2 * radius, xy[1] - 2 * radius
        k = 2 * t / radius ** 2 + 0.2 / radius
        return torch.stack([x + x ** 3 * k ** 2 / 6, y - x ** 2 * k / 2])
    return Twin(AutoBody(_plate(radius), bend),
                G.Body(G.Plate(radius - 2, 2.0), G.Bend2D(2 * radius, 2 / radius ** 2, 0.2 / radius)))
