"""Size-independent properties checked at the BASELINE sizes -- 256^3 Float32 (configs[1], C2) AND 512^3 Float32
(configs[2], C3: the headline size), both part of the default `-m gpu` run -- where the CPU oracle is too slow to be
the checker:

  * BC! is idempotent;  conv_diff! of a uniform stream is exactly zero on inside cells;
  * A is symmetric: x.(Ay) == y.(Ax);  mult! is exactly linear under power-of-two scaling;
  * restrict!(prolongate!(c)) == 8c (to the rounding of the partial sums);  restrict! conserves the sum;
  * solver! leaves r.r < tol and the projected velocity divergence-free to that tolerance;
  * an impulsively started uniform stream stays uniform (maintests.jl:172-180 at scale);
  * hydrostatic pressure_force on a sphere = its volume (maintests.jl:341-346 in 3-D)."""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SIZES = [int(v) for v in os.environ.get("WL_FULLSIZE", "256,512").split(",")]
T = np.float32


@pytest.fixture(scope="module")
def S():
    from waterlily_amd import sim
    return sim


@pytest.fixture(scope="module", params=SIZES, ids=[f"{n}^3" for n in SIZES])
def N(request):
    yield request.param
    import gc
    gc.collect()
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def flow(S, N):
    U = (2 / 3, -1 / 3, 0.25)
    a = S.Flow((N, N, N), U, T=T, ulam=lambda i, x: U[i])
    return a, S.MultiLevelPoisson(a.p, a.mu0, a.sigma), U


def rand_like(a, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.rand(a.shape, generator=g, device="cuda", dtype=a.dtype) - 0.5


def test_bc_idempotent_and_uniform_convdiff_zero(S, flow):
    a, ml, U = flow
    u1 = a.u.clone()
    S.BC(a.u, U)
    assert torch.equal(a.u, u1)
    S.conv_diff(a.f, a.u, nu=0.01)
    # (the two x-planes next to the exit are excluded: the Flow constructor's exitBC! (Flow.jl:115) shifts the exit
    #  plane by the rounding of its mean-flux correction)
    inner = (slice(1, -3),) + (slice(1, -1),) * 2
    assert float(a.f[inner].abs().max()) == 0.0


def test_operator_symmetry_and_linearity(S, flow):
    a, ml, U = flow
    lv = ml.levels[0]
    x, y = lv.layout.alloc((), "cuda:0"), lv.layout.alloc((), "cuda:0")
    inner = (slice(1, -1),) * 3
    x[inner] = rand_like(x[inner], 1)
    y[inner] = rand_like(y[inner], 2)
    Ay = S.copy_of(S.mult(ml, y))
    xAy = S.dot(x, Ay)
    Ax = S.copy_of(S.mult(ml, x))
    yAx = S.dot(y, Ax)
    assert abs(xAy - yAx) <= 1e-5 * max(abs(xAy), 1.0)
    x4 = S.like(x)
    x4[inner] = x[inner] * 4
    A4x = S.mult(ml, x4)
    assert torch.equal(A4x[inner], Ax[inner] * 4)


def test_restrict_prolongate_identities(S, flow):
    a, ml, U = flow
    f, c = ml.levels[0], ml.levels[1]
    ci = (slice(1, -1),) * 3
    c.x.zero_()
    c.x[ci] = rand_like(c.x[ci], 3)
    S.prolongate(f.eps, c.x)
    S.restrict(c.r, f.eps)
    # sum of the 8 identical children: 8c up to the roundings of the partial sums 3c,5c,6c,7c
    assert float((c.r[ci] - 8 * c.x[ci]).abs().max()) <= 4 * np.finfo(T).eps * 8 * float(c.x[ci].abs().max())
    f.r.zero_()
    f.r[ci] = rand_like(f.r[ci], 4)
    S.restrict(c.r, f.r)
    sf, sc = float(f.r.double().sum()), float(c.r.double().sum())
    assert abs(sf - sc) <= 1e-6 * max(1.0, float(f.r.double().abs().sum()))


def test_projection_divergence_free_and_uniform_stream(S, flow, N):
    a, ml, U = flow
    S.mom_step(a, ml)
    assert all(1 <= n <= 32 for n in ml.n[-2:])
    assert S.L2p(ml) < 1e-4                                   # solver! tolerance (MultiLevelPoisson.jl:87,95)
    z = S.like(a.p)
    S.divergence(z, a.u)
    assert S.L2(z) < 1e-3
    for i in range(3):                                        # impulsive uniform stream stays uniform
        d = (a.u[..., i] - U[i])[(slice(1, -1),) * 3]
        assert float((d.double() ** 2).sum()) < 2e-5 * (N / 16) ** 3


def test_hydrostatic_force_on_sphere(S, N):
    """maintests.jl:341-346 in 3-D at full size, through the whole product path: measure! on the device (band cells),
    the |d|<=1 band rebuilt from them, wl_pforce.  p = y  =>  force = volume * e_y."""
    import bodies
    R, c = N / 4, N / 2
    sim = S.Simulation((N, N, N), (1.0, 0.0, 0.0), 2 * R, body=bodies.sphere(c, R).product, T=T)
    yy = torch.arange(N + 2, device="cuda", dtype=torch.float32) - 0.5
    sim.flow.p.copy_(yy[None, :, None].expand(N + 2, N + 2, N + 2))
    force = S.pressure_force(sim)
    vol = 4 / 3 * math.pi * R ** 3
    assert np.sum(np.abs(force / vol - np.array([0, 1, 0]))) < 2e-3


def test_rows_per_thread_same_bits(S, flow, N):
    """wl_set_option(4): the 7-point kernel with one or two rows per thread evaluates the same per-cell expressions:
    mult!, Jacobi!+increment! and the fused V-cycle smoother give bit-identical fields at full size."""
    a, ml, U = flow
    lv = ml.levels[0]
    inner = (slice(1, -1),) * 3
    x = lv.layout.alloc((), "cuda:0")
    x[inner] = rand_like(x[inner], 11)
    r0 = S.like(x)
    r0[inner] = rand_like(r0[inner], 12)
    out = []
    for rows in (1, 2):
        S.set_option(4, rows)
        try:
            z = S.copy_of(S.mult(ml, x))
            lv.r.copy_(r0)
            a.p.zero_()
            S.Jacobi(ml)                                      # eps = r*iD ; r -= A eps ; x += eps
            out.append((z, S.copy_of(lv.r), S.copy_of(a.p)))
        finally:
            S.set_option(4, 0)
    for u, v in zip(*out):
        assert torch.equal(u, v)


def test_traffic_saving_switches_do_not_change_a_bit(S, N):
    """The kernels that move fewer bytes than the dense algorithm -- row constants instead of L/iD in coefficient-
    uniform rows (option 9), x += alpha*eps deferred to the direction kernel (8), z' = r*iD recomputed instead of
    stored (13), z = A*eps formed a second time by the update kernel instead of stored (19), body-free rows in BDIM! (3), the chained x/=dt ; x*=dt' pass (14), the shared-flux conv_diff! kernel (18, 20), its x-ghost launch (21), div(u)
    formed inside residual! (22), the x planes of BC! written by the producing kernel (23) -- evaluate the same expressions: three steps of the sphere case give
    bit-identical u and p with all of them off."""
    import bench
    sims = []
    for on in (1, 0):
        for key in (3, 8, 9, 13, 14, 18, 19, 20, 21, 22, 23):
            S.set_option(key, (2 if on else 0) if key == 19 else on)     # (19: 2 = on every level, also the 512^3 one)
        try:
            sim = bench.sphere((N, N, N), T)
            for _ in range(3):
                S.sim_step(sim, remeasure=False)
        finally:
            for key in (3, 8, 9, 13, 14, 18, 19, 20, 21, 22, 23):
                S.set_option(key, 1)
        sims.append(sim)
    a, b = sims
    nu, nr = S.uniform_rows(a.pois, 0)
    assert nu > 0.8 * nr                      # most rows of the sphere case are coefficient-uniform
    assert a.pois.n == b.pois.n and a.flow.dt == b.flow.dt
    assert torch.equal(a.flow.u, b.flow.u) and torch.equal(a.flow.p, b.flow.p)
