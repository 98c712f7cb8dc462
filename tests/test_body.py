"""Pins for the host-side geometry module against the reference's own known-answer tests
(/root/reference/test/maintests.jl:183-230, "Body.jl" and "AutoBody.jl" test sets)."""
import math

import numpy as np
import torch

from waterlily_amd import body as B
from waterlily_amd.body import AutoBody, measure, norm2

S2, SH = math.sqrt(2.0), math.sqrt(0.5)


def _close(got, want, tol=1e-12):
    for g, w in zip(got, want):
        assert np.allclose(np.asarray(g, dtype=float), np.asarray(w, dtype=float), atol=tol), (got, want)


def test_kernel_moments():  # maintests.jl:184-186
    assert B.mu0(3, 6) == B.mu0(0.5, 1)
    assert B.mu0(0, 1) == 0.5
    assert B.mu1(0, 2) == 2 * (1 / 4 - 1 / math.pi ** 2)


def test_measure_autodiff_2d_3d():  # maintests.jl:192-197
    body1 = AutoBody(lambda x, t: norm2(x) - 2 - t)
    _close(measure(body1, [S2, S2], 0.0), (0, [SH, SH], [0, 0]))
    _close(measure(body1, [2.0, 0.0, 0.0], 1.0), (-1.0, [1, 0, 0], [0, 0, 0]))
    body2 = AutoBody(lambda x, t: norm2(x) - 2, lambda x, t: x + t ** 2)
    _close(measure(body2, [S2, S2], 0.0), (0, [SH, SH], [0, 0]))
    _close(measure(body2, [1.0, -1.0, -1.0], 1.0), (0.0, [1, 0, 0], [-2, -2, -2]))


def test_booleans():  # maintests.jl:199-202
    body1 = AutoBody(lambda x, t: norm2(x) - 2 - t)
    body2 = AutoBody(lambda x, t: norm2(x) - 2, lambda x, t: x + t ** 2)
    _close(measure(body1 + body2, [-S2, -S2], 1.0), (-S2, [-SH, -SH], [-2, -2]))
    _close(measure(body1 | body2, [-S2, -S2], 1.0), (-S2, [-SH, -SH], [-2, -2]))
    _close(measure(body1 - body2, [-S2, -S2], 1.0), (S2, [SH, SH], [-2, -2]))


def test_fast_version():  # maintests.jl:227-229
    body1 = AutoBody(lambda x, t: norm2(x) - 2 - t)
    _close(measure(body1, [3.0, 4.0], 0.0, fastd2=9), measure(body1, [3.0, 4.0], 0.0))
    d, n, V = measure(body1, [3.0, 4.0], 0.0, fastd2=8)
    _close((d, n, V), (B.sdf(body1, [3.0, 4.0], 0.0), [0, 0], [0, 0]))


def test_measure_sdf_matches_closure():  # maintests.jl:221-225
    body1 = AutoBody(lambda x, t: norm2(x) - 2 - t)
    _, _, _, d = B.measure_fields(body1, (2, 3), t=0.0, eps=1, T=np.float32)
    # cell I=(2,3) (1-based) -> python (1,2); loc(0,I) = I-1.5
    x = torch.tensor([0.5, 1.5], dtype=torch.float64)
    assert abs(float(d[1, 2]) - float(B.sdf(body1, x, 0.0))) < 1e-6


def test_measure_fields_circle_properties():
    """Body.jl:31-50: mu0 in [0,1], 0 deep inside, 1 far outside; mu1 = eps*kern1*n; V=0 for a static body."""
    R, c = 8.0, 15.0
    body = AutoBody(lambda x, t: norm2(x - c) - R)
    m0, m1, V, d = B.measure_fields(body, (32, 32), T=np.float64)
    assert abs(m0.min()) < 1e-16 and m0.max() == 1.0  # kern0(-1) = -1.9e-17 in Float64, as in Julia
    assert np.all(V == 0)
    # centre cell is deep inside -> mu0 = 0 ; corner far outside -> 1
    assert np.all(np.abs(m0[16, 16]) < 1e-16) and np.all(m0[2, 2] == 1)
    # face (i=0) of a cell cut by the surface: compare with the closed form
    I = (int(c + R + 0.5) + 1, 16)          # x-face near the +x pole
    xf = np.array([I[0] - 0.5 - 0.5, I[1] - 0.5])
    dist = np.hypot(*(xf - c)) - R
    assert abs(m0[I + (0,)] - B.mu0(dist, 1)) < 1e-12
    nrm = (xf - c) / np.hypot(*(xf - c))
    assert np.allclose(m1[I + (0,)], B.mu1(dist, 1) * nrm, atol=1e-12)


def test_moving_body_velocity():
    """V = -J^-1 dmap/dt (AutoBody.jl:124-130): translating circle x - [t,0] moves with V=(1,0)."""
    r = 8.0
    circle = lambda x, t: norm2(x - 2 * r) - r
    move = lambda x, t: x - torch.stack([t, torch.zeros_like(t)])[:, None]
    body = AutoBody(circle, move)
    d, n, V = measure(body, [2 * r + r, 2 * r], 0.0)
    _close((d, n, V), (0.0, [1, 0], [1, 0]))


def test_nds_band_hydrostatic_2d():  # maintests.jl:341-346 geometry: integral of y*n*kern = area*e_y
    N = 32
    body = AutoBody(lambda x, t: norm2(x - N / 2) - N // 4)
    idx, nds = B.nds_band(body, (N - 2, N - 2))
    Ng = (N, N)
    j = idx // Ng[0]
    y = j - 0.5
    force = (y[:, None] * nds).sum(0)
    assert np.sum(np.abs(force / (math.pi * (N / 4) ** 2) - np.array([0, 1]))) < 2e-3


def test_set_operations_on_parametric_bodies_keep_their_grouping():
    """ParametricBody / Bodies operators: only a union of unions may be flattened into one left fold of the leaves;
    a + (b - c), a | (b & c), a - (b + c), a & (b + c) must equal the nested AutoBody closures (AutoBody.jl:22-34)."""
    D = 2
    a, b, c = B.Sphere((10.0, 10.0), 4.0, D), B.Sphere((14.0, 10.0), 4.0, D), B.Sphere((8.0, 10.0), 3.0, D)
    ca, cb, cc = (AutoBody(q.sdf) for q in (a, b, c))      # the same leaves as plain closures: operators nest
    pts = torch.tensor([[11.0, 7.0, 9.0, 13.0, 16.0, 4.5], [10.0, 10.0, 11.5, 12.0, 10.0, 10.0]], dtype=torch.float64)
    cases = [
        (a + (b - c), ca + (cb - cc), False),
        (a | (b & c), ca | (cb & cc), False),
        (a - (b + c), ca - (cb + cc), False),
        (a & (b + c), ca & (cb + cc), False),
        (a + (b + c), ca + (cb + cc), True),     # union of unions: one native composite of three leaves
        ((a - b) + c, (ca - cb) + cc, True),     # a left fold as written
    ]
    for got, want, native in cases:
        assert np.allclose(B.sdf(got, pts).numpy(), B.sdf(want, pts).numpy(), atol=0, rtol=0)
        assert (isinstance(got, B.Bodies) and B.is_native(got)) == native
    # the advisor's two points: inside a ∩ c the wrong left folds (a ∪ b) − c and (a ∪ b) ∩ c differ from the nested bodies
    assert float(B.sdf(a + (b - c), [11.0, 10.0])) == -3.0 and float(B.sdf(a | (b & c), [7.0, 10.0])) == -1.0
    # a composite that stays native is described leaf by leaf, in order
    d3 = (a + (b + c)).native_desc(0.0, D)
    assert d3[0].count == 3 and [d3[l].op for l in range(3)] == [0, 0, 0]
