"""Pins the CPU oracle (oracle/) with the reference's OWN known-answer and analytic tests.

Each test restates one test of /root/reference/test/maintests.jl (line ranges cited); the reference ships
no golden files, so these analytic pins are what anchors the oracle.  Indices are
0-based here (reference index - 1); periodic directions are 0-based too.
"""
import math

import numpy as np
import pytest

from oracle import geometry as G
from oracle import wl_oracle as O


def sim(*a, **k):
    """bodies are oracle.geometry.Body objects (closed-form sdf/map, numpy): no product code feeds these pins"""
    return O.Simulation(*a, **k)


# ----------------------------------------------------------------------------- util.jl (maintests.jl:5-66)

def test_loc():  # :12-14
    x = O.loc(2, (6, 6, 6))[:, 2, 3, 4]           # loc(3, CI(3,4,5))
    assert np.all(x == np.array([3, 4, 4.5]) - 1.5)
    I = (4, 7, 2)
    assert np.all(O.loc(-1, (11, 11, 11))[(slice(None),) + I] == np.array(I) + 1 - 1.5)


@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_apply_inside_L2(T):  # :22-30
    p = O.zeros((4, 5), T)
    O.apply_scalar(lambda x: x[0] + x[1] + 3, p)
    assert O.inside(p) == (slice(1, 3), slice(1, 4))
    assert O.L2(p) == 187
    u = O.zeros((5, 5, 2), T)
    O.apply_vec(lambda i, x: x[i], u)
    assert all(u[i, j, 0] - (i + 1 - 2) == 0 for i in range(3) for j in range(3))


@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_BC_exitBC_perBC(T):  # :32-56
    rng = np.random.default_rng(0)
    Ng, D, U = (6, 6), 2, (1.0, 0.5)
    u = np.asfortranarray(rng.random(Ng + (D,)).astype(T))
    s = np.asfortranarray(rng.random(Ng).astype(T))
    O.BC(u, U)
    assert np.all(u[0, :, 0] == U[0]) and np.all(u[1, :, 0] == U[0]) and np.all(u[-1, :, 0] == U[0])
    assert np.all(u[2:-1, 0, 0] == u[2:-1, 1, 0]) and np.all(u[2:-1, -1, 0] == u[2:-1, -2, 0])
    assert np.all(u[:, 0, 1] == U[1]) and np.all(u[:, 1, 1] == U[1]) and np.all(u[:, -1, 1] == U[1])
    assert np.all(u[0, 2:-1, 1] == u[1, 2:-1, 1]) and np.all(u[-1, 2:-1, 1] == u[-2, 2:-1, 1])

    u[-1, :, 0] = 3
    O.BC(u, U, True)                                  # save exit values
    assert np.all(u[-1, :, 0] == 3)
    O.exitBC(u, u, U, 0)                              # conservative exit check
    assert np.all(u[-1, 1:-1, 0] == U[0])

    O.BC(u, U, True, (1,))                            # periodic in y and save exit values
    assert np.all(u[:, 0:2, 0] == u[:, -2:, 0])
    O.perBC(s, (0, 1))
    assert np.all(s[0, 1:-1] == s[-2, 1:-1]) and np.all(s[1:-1, 0] == s[1:-1, -2])

    u = np.asfortranarray(rng.random(Ng + (D,)).astype(T))
    O.BC(u, U, True, (0,))                            # saveexit has no effect: x-periodic
    assert np.all(u[0:2, :, 0] == u[-2:, :, 0]) and np.all(u[0:2, :, 1] == u[-2:, :, 1])
    assert np.all(u[:, 0, 1] == U[1]) and np.all(u[:, 1, 1] == U[1]) and np.all(u[:, -1, 1] == U[1])


# ----------------------------------------------------------------------------- Poisson.jl (maintests.jl:68-94)

def Poisson_setup(poisson, N, T=np.float32):
    D = len(N)
    c = np.ones(N + (D,), dtype=T, order="F")
    O.BC(c, (0.0,) * D)
    x = O.zeros(N, T)
    z = O.zeros(N, T)
    pois = poisson(x, c, z)
    soln = np.asfortranarray(np.broadcast_to(
        np.arange(1, N[0] + 1, dtype=T).reshape((-1,) + (1,) * (D - 1)), N).copy())
    I = (1,) * D
    soln -= soln[I]
    O.mult(pois, soln)
    O.solver(pois)
    x -= x[I]
    return O.L2(x - soln) / O.L2(soln), pois


def test_poisson_diag_5x5():  # :83-86
    err, pois = Poisson_setup(O.Poisson, (5, 5))
    D = np.array([[0, 0, 0, 0, 0], [0, -2, -3, -2, 0], [0, -3, -4, -3, 0], [0, -2, -3, -2, 0], [0, 0, 0, 0, 0]], np.float32)
    assert np.array_equal(pois.D, D)
    with np.errstate(divide="ignore"):
        iD = np.where(D == 0, 0, 1 / D).astype(np.float32)
    assert np.allclose(pois.iD, iD)
    assert err < 1e-5


def test_poisson_single_level_iterations():  # :87-92
    err, pois = Poisson_setup(O.Poisson, (2 ** 6 + 2, 2 ** 6 + 2))
    assert err < 1e-6 and pois.n[0] < 310
    err, pois = Poisson_setup(O.Poisson, (2 ** 4 + 2,) * 3)
    assert err < 1e-6 and pois.n[0] < 35


# ----------------------------------------------------------------------------- MultiLevelPoisson.jl (:96-117)

def test_up_down_inverse():  # :97-98 -- via restrict/prolongate index maps
    fine = O.zeros((10, 8, 6), np.float64)
    coarse = O.zeros((6, 5, 4), np.float64)
    coarse[...] = np.arange(coarse.size, dtype=np.float64).reshape(coarse.shape, order="F")
    O.prolongate(fine, coarse)
    # every child J in up(I) must map back to I: children of coarse (3,2,1) are fine {5,6}x{3,4}x{1,2}
    assert np.all(fine[5:7, 3:5, 1:3] == coarse[3, 2, 1])
    back = O.zeros(coarse.shape, np.float64)
    O.restrict(back, fine)
    assert np.all(back[O.inside(back)] == 8 * coarse[O.inside(coarse)])


def test_mlp_size_assertion():  # :99
    with pytest.raises(AssertionError, match="MultiLevelPoisson requires size=a2ⁿ, where n>2"):
        Poisson_setup(O.MultiLevelPoisson, (15 + 2, 3 ** 4 + 2))


def test_mlp_coarse_diag_and_update():  # :101-107
    err, pois = Poisson_setup(O.MultiLevelPoisson, (10, 10))
    assert np.array_equal(pois.levels[2].D, np.array([[0, 0, 0, 0], [0, -2, -2, 0], [0, -2, -2, 0], [0, 0, 0, 0]], np.float32))
    assert err < 1e-5
    pois.levels[0].L[4:6, :, 0] = 0
    O.update(pois)
    assert np.array_equal(pois.levels[2].D, np.array([[0, 0, 0, 0], [0, -1, -1, 0], [0, -1, -1, 0], [0, 0, 0, 0]], np.float32))


@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_mlp_iterations(T):  # :109-116
    err, pois = Poisson_setup(O.MultiLevelPoisson, (2 ** 6 + 2, 2 ** 6 + 2), T)
    assert err < 1e-6 and pois.n[0] <= 3
    err, pois = Poisson_setup(O.MultiLevelPoisson, (2 ** 4 + 2,) * 3, T)
    assert err < 1e-6 and pois.n[0] <= 3


# ----------------------------------------------------------------------------- Flow.jl (:119-181)

def test_vanLeer():  # :121-123
    assert O.vanLeer(1, 0, 1) == 0 and O.vanLeer(1, 2, 1) == 2
    assert O.vanLeer(1, 2, 3) == 2.5 and O.vanLeer(3, 2, 1) == 1.5


def test_quick_boundary_fluxes():  # :125-138  (python index = reference index - 1)
    f = [0.0, 0.5, 2.0]
    assert O.phiuL(1, f, 1) == O.phi(1, f)                     # inlet, positive flux -> CD
    assert O.phiuL(1, f, -1) == -O.quick(2.0, 0.5, 0.0)        # inlet, negative flux -> backward QUICK
    assert O.phiuR(2, f, 1) == O.quick(0.0, 0.5, 2.0)          # outlet, positive flux -> QUICK
    assert O.phiuR(2, f, -1) == -O.phi(2, f)                   # outlet, negative flux -> backward CD


def test_phiu_phiuP():  # :140-155
    f = [0.0, 0.5, 2.0]
    assert O.phiu(2, f, 1) == O.phiuP(0, 2, f, 1)
    f4 = [0.0, 0.5, 2.0, 0.0]                                   # room for the I+1 read of the negative branch
    assert O.phiu(1, f4, -1) == O.phiuP(-1 % 4, 1, f4, -1)
    f = [1.0, 1.25, 1.5, 1.75, 2.0]
    assert O.phiuP(0, 2, f, 1) == O.quick(f[0], f[1], f[2])
    Ip = len(f) - 2 - 1                                         # CIj(1,I,length(f)-2)
    assert O.phiuP(Ip, 2, f, 1) == O.quick(f[Ip], f[1], f[2])


def test_BCTuple():  # :157-158
    assert O.BCTuple((1, 2, 3), [0], 3) == O.BCTuple(lambda i, t: i + 1, [0], 3)
    assert O.BCTuple(lambda i, t: t, [1.234], 3) == (1.234,) * 3


@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_accelerate(T):  # :161-171
    N = 4
    a = O.zeros((N, N, 2), T)
    assert O.accel_tuple(None, (), [1], 2) is None
    O.accelerate(a, O.accel_tuple(lambda i, t: t if i == 0 else 2 * t, (), [1], 2))
    assert np.all(a[:, :, 0] == 1) and np.all(a[:, :, 1] == 2)
    O.accelerate(a, O.accel_tuple(None, lambda i, t: -t if i == 0 else -2 * t, [1], 2))
    assert np.allclose(a, 0, atol=1e-6)
    O.accelerate(a, O.accel_tuple(lambda i, t: t if i == 0 else 2 * t, lambda i, t: -t if i == 0 else -2 * t, [1], 2))
    assert np.allclose(a, 0, atol=1e-6)


def test_impulsive_flow_in_box():  # :172-180
    U = (2 / 3, -1 / 3)
    N = (2 ** 4, 2 ** 4)
    a = O.Flow(N, U, T=np.float32)
    O.mom_step(a, O.MultiLevelPoisson(a.p, a.mu0, a.sigma))
    assert O.L2(a.u[:, :, 0] - np.float32(U[0])) < 2e-5
    assert O.L2(a.u[:, :, 1] - np.float32(U[1])) < 1e-5


# ----------------------------------------------------------------------------- periodic TGV (:232-253)

def TGV(i, xy, t, k, nu):
    x, y = xy[0] * k, xy[1] * k
    if i == 0:
        return -np.sin(x) * np.cos(y) * np.exp(-2 * k ** 2 * nu * t)
    return np.cos(x) * np.sin(y) * np.exp(-2 * k ** 2 * nu * t)


def TGVsim(Re=1e8, T=np.float64):
    L = 64
    k = 2 * math.pi / L
    nu = 1 / (k * Re)
    return sim((L, L), (0, 0), L, U=1, ulam=lambda i, x: TGV(i, x, 0.0, k, nu), nu=nu, T=T, perdir=(0, 1)), k, nu


def test_periodic_TGV():
    s, k, nu = TGVsim(T=np.float32)
    ue = s.flow.u.copy(order="F")
    O.sim_step(s, math.pi / 100)
    O.apply_vec(lambda i, x: TGV(i, x, O.time(s.flow), 2 * math.pi / s.L, s.flow.nu), ue)
    u = s.flow.u
    assert O.L2(u[:, :, 0] - ue[:, :, 0]) < 1e-4 and O.L2(u[:, :, 1] - ue[:, :, 1]) < 1e-4


# ----------------------------------------------------------------------------- accelerating flow (:280-302)

def test_flow_with_increasing_body_force():
    N, jerk = 8, 4
    UScale = math.sqrt(N)
    s = sim((N, N), (UScale, 0.0), N, nu=0.001, g=lambda i, t: t * jerk if i == 0 else 0.0, dt=0.001,
            perdir=(0,), T=np.float64)
    O.sim_step(s, 1.0)
    u = s.flow.u
    uFinal = s.flow.U[0] + 0.5 * jerk * O.time(s.flow) ** 2
    assert O.L2(u[:, :, 0] - uFinal) < 1e-4 and O.L2(u[:, :, 1] - 0) < 1e-4


# ----------------------------------------------------------------------------- accelerating circle (:304-316)

@pytest.mark.slow
def test_circle_in_accelerating_flow():
    radius, H = 32, 16
    c = float(H * radius)
    s = sim((radius * 2 * H, radius * 2 * H), lambda i, t: t if i == 0 else 0.0 * t, radius, U=1,
            body=G.Body(G.Sphere(c, radius)))
    O.sim_step(s)
    f = O.pressure_force(s) / (math.pi * s.L ** 2)
    assert np.allclose(f, [-1, 0], atol=0.04)
    assert s.flow.u.max() / s.flow.u[1, 1, 0] > 1.91         # ~2U
    for _ in range(3):
        O.sim_step(s)
    assert all(n <= 2 for n in s.pois.n)


# ----------------------------------------------------------------------------- Metrics.jl hydrostatic force (:341-346)

@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_hydrostatic_pressure_force(T):
    N = 32
    p = O.zeros((N, N), T)
    p[O.inside(p)] = O.loc(-1, (N, N))[1][O.inside(p)].astype(T)
    df = O.zeros((N, N, 2), T)
    body = G.Body(G.Sphere(N / 2, N // 4))
    idx, nds = G.nds_band(body, (N - 2, N - 2))
    force = O.pressure_force_band(p, df, idx, nds)
    assert np.sum(np.abs(force / (math.pi * (N / 4) ** 2) - np.array([0, 1]))) < 2e-3


# ----------------------------------------------------------------------------- WaterLily.jl (:372-413)

RADIUS = 8
NU = RADIUS / 250
NM = (RADIUS * 4, RADIUS * 4)


_circle = G.Sphere(2.0 * RADIUS, RADIUS)                                       # :372
_plate = G.Plate(RADIUS - 2.0, 2.0)                                            # :375


def test_sim_time_stopping():  # :387-390
    s = sim(NM, (1, 0), RADIUS, body=G.Body(_circle), nu=NU, T=np.float32)
    assert O.sim_time(s) == 0
    O.sim_step(s, 0.1, remeasure=False)
    assert O.sim_time(s) >= 0.1 > sum(s.flow.dt[:-2]) * s.U / s.L


@pytest.mark.parametrize("exitBC", [True, False])
def test_moving_bodies(exitBC):  # :391-412
    kw = dict(nu=NU, T=np.float32, exitBC=exitBC)
    # remeasure works perfectly when V = U = 1
    s = sim(NM, (1, 0), RADIUS, body=G.Body(_circle, G.Translate(v=(1.0, 0.0))), **kw)           # move, :373
    O.sim_step(s)
    assert np.allclose(s.flow.u[:, RADIUS - 1, 0], 1, rtol=1e-6 ** 0.5)   # Julia's Float32 `≈`
    # accelerating from U=0 to U=1
    s = sim(NM, (0, 0), RADIUS, U=1, body=G.Body(_circle, G.Translate(a=(2.0, 0.0))), **kw)      # accel, :374
    O.sim_step(s)
    assert s.pois.n == [2, 1]
    assert s.flow.u.max() > s.flow.V.max() > 0
    # non-uniform V doesn't break
    s = sim(NM, (0, 0), RADIUS, U=1, body=G.Body(_plate, G.Rotate2D(2.0 * RADIUS, 1 / RADIUS, 1.0)), **kw)   # :376-379
    O.sim_step(s)
    assert s.pois.n == [2, 1]
    assert 1 > s.flow.dt[-1] > 0.5
    # divergent V doesn't break
    s = sim(NM, (0, 0), RADIUS, U=1, body=G.Body(_plate, G.Bend2D(2.0 * RADIUS, 2 / RADIUS ** 2, 0.2 / RADIUS)), **kw)   # :380-383
    O.sim_step(s)
    assert s.pois.n == [2, 1]
    assert 1.2 > s.flow.dt[-1] > 0.8


# ----------------------------------------------------------------------------- Metrics.jl viscous force / moment (:360-368)

def test_viscous_force_and_pressure_moment():
    N = 32
    for D in (2, 3):
        shp = (N,) * D
        body = G.Body(G.Sphere(N / 2, N // 4))
        idx, nds = G.nds_band(body, tuple(n - 2 for n in shp))
        u = O.zeros(shp + (D,), np.float64)
        df = O.zeros(shp + (D,), np.float64)
        assert np.allclose(O.viscous_force_band(u, 1.0, df, idx, nds), 0)                       # :362-363
        O.apply_vec(lambda i, x: x[i], u)                                                        # uniform dilatation:
        f = O.viscous_force_band(u, 1.0, df, idx, nds)                                           # -nu*2I*oint(n ds) = 0
        assert np.allclose(f, 0, atol=1e-9)
        p = O.zeros(shp, np.float64)
        O.apply_scalar(lambda x: x[1], p)
        m = O.pressure_moment_band((N / 2,) * D, p, df, idx, nds)                                # :365-368
        assert np.allclose(m, 0, atol=1e-8 * N ** D)


def test_field_metrics():  # maintests.jl:319-339
    J = (1, 2, 3)                                   # CartesianIndex(2,3,4)
    x = np.array(J) + 1 - 1.5
    px = np.prod(x)
    u = O.zeros((3, 4, 5, 3), np.float64)
    O.apply_vec(lambda i, xx: xx[i] + xx[0] * xx[1] * xx[2], u)
    p = O.zeros((3, 4, 5), np.float64)
    assert O.metric(p, "ke", u)[J] == 0.5 * np.sum((x + px) ** 2)
    assert O.metric(p, "ke", u, par=x)[J] == 1.5 * px ** 2
    assert abs(O.metric(p, "lambda2", u)[J] - 1) < 1e-12
    w = np.cross(1 / x, np.repeat(px, 3))
    assert O.metric(p, "curl", u, i=1)[J] == w[1]
    assert abs(O.metric(p, "omega_mag", u)[J] - np.sqrt(np.sum(w ** 2))) < 1e-12
    assert abs(O.metric(p, "omega_theta", u, par=(0, 0, 1), par2=x + np.array([0, 1, 2]))[J] - w[0]) < 1e-12
