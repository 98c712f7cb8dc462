"""oracle/geometry.py (closed-form body measurement) pinned by the reference's own known answers
(/root/reference/test/maintests.jl:183-206,221-229), then used as the INDEPENDENT checker of the product's host-side
geometry (waterlily_amd.body: torch + autograd) on every body family of the BASELINE configs and of the reference's
moving-body tests (maintests.jl:371-384).  CPU only."""
import math

import numpy as np
import pytest
import torch

from oracle import geometry as G
from waterlily_amd import body as B
from waterlily_amd.body import AutoBody, norm2

S2, SH = math.sqrt(2.0), math.sqrt(0.5)


def _close(got, want, tol=1e-12):
    for g, w in zip(got, want):
        assert np.allclose(np.asarray(g, dtype=float), np.asarray(w, dtype=float), atol=tol), (got, want)


# ------------------------------------------------------------------ pins
def test_kernel_moments():  # maintests.jl:184-186
    assert G.mu0(3, 6) == G.mu0(0.5, 1)
    assert G.mu0(0, 1) == 0.5
    assert G.mu1(0, 2) == 2 * (1 / 4 - 1 / math.pi ** 2)


def test_measure_known_answers():  # maintests.jl:192-197
    body1 = G.Body(G.Sphere(0.0, 2.0, growth=1.0))                       # norm2(x)-2-t
    _close(G.measure(body1, [S2, S2], 0.0), (0, [SH, SH], [0, 0]))
    _close(G.measure(body1, [2.0, 0.0, 0.0], 1.0), (-1.0, [1, 0, 0], [0, 0, 0]))
    body2 = G.Body(G.Sphere(0.0, 2.0), G.Translate(a=-1.0))               # map x .+ t^2
    _close(G.measure(body2, [S2, S2], 0.0), (0, [SH, SH], [0, 0]))
    _close(G.measure(body2, [1.0, -1.0, -1.0], 1.0), (0.0, [1, 0, 0], [-2, -2, -2]))


def test_fast_version():  # maintests.jl:227-229
    body1 = G.Body(G.Sphere(0.0, 2.0, growth=1.0))
    _close(G.measure(body1, [3.0, 4.0], 0.0, fastd2=9), G.measure(body1, [3.0, 4.0], 0.0))
    _close(G.measure(body1, [3.0, 4.0], 0.0, fastd2=8), (G.sdf(body1, [3.0, 4.0], 0.0), [0, 0], [0, 0]))


def test_measure_sdf_matches_closed_form():  # maintests.jl:221-225
    body1 = G.Body(G.Sphere(0.0, 2.0, growth=1.0))
    _, _, _, d = G.measure_fields(body1, (2, 3), t=0.0, eps=1, T=np.float32)
    assert abs(float(d[1, 2]) - (math.hypot(0.5, 1.5) - 2)) < 1e-6         # I=(2,3): loc(0,I) = I-1.5


def test_hydrostatic_band():  # maintests.jl:341-346: integral of y*n*kern over the band = area*e_y
    N = 32
    body = G.Body(G.Sphere(N / 2, N // 4))
    idx, nds = G.nds_band(body, (N - 2, N - 2))
    y = idx // N - 0.5
    force = (y[:, None] * nds).sum(0)
    assert np.sum(np.abs(force / (math.pi * (N / 4) ** 2) - np.array([0, 1]))) < 2e-3


# ------------------------------------------------------------------ product (torch closures + autograd) vs oracle (closed forms)
def _vec(t, *comps):
    return torch.stack([c if isinstance(c, torch.Tensor) else torch.full_like(t, c) for c in comps])[:, None]


def _cases():
    r = 8.0
    circle = lambda x, t: norm2(x - 2 * r) - r
    plate = lambda x, t: norm2(x - torch.stack([torch.clamp(x[0], -r + 2, r - 2), torch.zeros_like(x[0])])) - 2

    def rotate(x, t):
        s, c = torch.sin(t / r + 1), torch.cos(t / r + 1)
        e = x - 2 * r
        return torch.stack([c * e[0] + s * e[1], -s * e[0] + c * e[1]])

    def bend(xy, t):
        x, y = xy[0] - 2 * r, xy[1] - 2 * r
        k = 2 * t / r ** 2 + 0.2 / r
        return torch.stack([x + x ** 3 * k ** 2 / 6, y - x ** 2 * k / 2])

    def torus(x, t, c=16.0, R=8.0, rr=2.0):
        ring = torch.sqrt((x[1] - c) ** 2 + (x[2] - c) ** 2) - R
        return torch.sqrt((x[0] - c) ** 2 + ring ** 2) - rr
    return {
        # name: (dims, product body, oracle body, times)
        "circle": ((48, 16), AutoBody(lambda x, t: norm2(x - 7.0) - 2.0), G.Body(G.Sphere(7.0, 2.0)), (0.0,)),
        "sphere": ((16, 16, 16), AutoBody(lambda x, t: norm2(x - 7.0) - 2.0), G.Body(G.Sphere(7.0, 2.0)), (0.0,)),
        "torus": ((32, 32, 32), AutoBody(torus), G.Body(G.Torus(16.0, 8.0, 2.0)), (0.0,)),
        "move": ((32, 32), AutoBody(circle, lambda x, t: x - _vec(t, t, 0.0)),
                 G.Body(G.Sphere(2 * r, r), G.Translate(v=(1.0, 0.0))), (0.0, 0.75)),
        "accel": ((32, 32), AutoBody(circle, lambda x, t: x - _vec(t, 2 * t ** 2, 0.0)),
                  G.Body(G.Sphere(2 * r, r), G.Translate(a=(2.0, 0.0))), (0.0, 0.5)),
        "plate_rotate": ((32, 32), AutoBody(plate, rotate), G.Body(G.Plate(r - 2, 2.0), G.Rotate2D(2 * r, 1 / r, 1.0)), (0.0, 0.6)),
        "plate_bend": ((32, 32), AutoBody(plate, bend), G.Body(G.Plate(r - 2, 2.0), G.Bend2D(2 * r, 2 / r ** 2, 0.2 / r)), (0.0, 0.6)),
        # the 3-D cylinder of the reference's examples (norm2(x[1:2] .- c) - R) and AutoBody set operations / `Bodies`
        # (AutoBody.jl:22-34,40-110): the product side as closures combined with +, -, ∩; the oracle side as G.Bodies
        "cylinder": ((24, 24, 12), B.Cylinder(11.0, 4.0, 3), G.Body(G.Cylinder(11.0, 4.0, (0, 1))), (0.0,)),
        "union_move": ((40, 24), AutoBody(lambda x, t: norm2(x - 9.0) - 4.0) + AutoBody(lambda x, t: norm2(x - 12.3) - 3.1, lambda x, t: x - _vec(t, 1.5 * t, 0.0)),
                       G.Bodies([G.Body(G.Sphere(9.0, 4.0)), G.Body(G.Sphere(12.3, 3.1), G.Translate(v=(1.5, 0.0)))], ["+"]), (0.0, 2.0)),   # (no exact ties between the leaves)
        "sphere_minus_cyl_and": ((24, 24, 24), (B.Sphere(11.0, 7.0, 3) - B.Cylinder(11.0, 3.0, 3)) & B.Sphere((11.0, 11.0, 14.0), 7.0, 3),
                                 G.Bodies([G.Body(G.Sphere(11.0, 7.0)), G.Body(G.Cylinder(11.0, 3.0, (0, 1))), G.Body(G.Sphere((11.0, 11.0, 14.0), 7.0))],
                                          ["-", "&"]), (0.0,)),
    }


@pytest.mark.parametrize("name", list(_cases()))
@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_product_host_geometry_matches_oracle(name, T):
    """mu0, mu1, V, sigma of Body.jl:31-50 from waterlily_amd.body (autograd) == closed-form oracle, to rounding:
    both evaluate in Float64 and round to T, so all but a handful of entries are bit-identical and none differs by
    more than a few ulp (autograd and the analytic gradient differ in the last Float64 bits)."""
    dims, pb, ob, times = _cases()[name]
    tol = 4 * np.finfo(T).eps
    for t in times:
        got = B.measure_fields(pb, dims, t=t, eps=1, T=T)
        want = G.measure_fields(ob, dims, t=t, eps=1, T=T)
        for g, w, nm in zip(got, want, ("mu0", "mu1", "V", "d")):
            assert g.shape == w.shape and g.dtype == w.dtype
            scale = max(1.0, float(np.abs(w).max()))
            assert np.abs(g.astype(np.float64) - w).max() <= tol * scale, (name, nm, t)
        assert np.abs(want[1]).max() > 0.1                                 # the case does exercise the band ...
        assert name in ("circle", "sphere", "torus", "cylinder", "sphere_minus_cyl_and") or t == 0.0 or np.abs(want[2]).max() > 0.1   # ... and the body velocity


@pytest.mark.parametrize("name", ["circle", "sphere", "torus", "accel"])
def test_product_nds_band_matches_oracle(name):
    dims, pb, ob, times = _cases()[name]
    t = times[-1]
    i1, v1 = B.nds_band(pb, dims, t=t)
    i2, v2 = G.nds_band(ob, dims, t=t)
    assert np.array_equal(i1, i2)
    assert np.abs(v1 - v2).max() < 1e-13
