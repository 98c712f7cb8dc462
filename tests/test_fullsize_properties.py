"""Size-independent properties checked at the BASELINE sizes, where the CPU oracle is too slow to be the checker -- all
part of the default `-m gpu` run, each on ONE GPU:

  C2  256^3            Float32 sphere   (configs[1])
  C3  512^3            Float32 sphere   (configs[2]: the headline size)
  C4  1024x1024x512    Float32 sphere   (configs[3]: the 8-GPU strong-scaling grid; 67.7 GB of fields and levels)
  C5  512^3            Float64 torus    (configs[4]: the double-precision path)

  * BC! is idempotent;  conv_diff! of a uniform stream is exactly zero on inside cells;
  * A is symmetric: x.(Ay) == y.(Ax);  mult! is exactly linear under power-of-two scaling;
  * restrict!(prolongate!(c)) == 8c (to the rounding of the partial sums);  restrict! conserves the sum;
  * solver! leaves r.r < tol and the projected velocity divergence-free to that tolerance;
  * an impulsively started uniform stream stays uniform (maintests.jl:172-180 at scale);
  * hydrostatic pressure_force on the body = its volume (maintests.jl:341-346 in 3-D);
  * the traffic-saving kernel forms and the rows-per-thread variants do not change a bit;
  * two steps of the configuration itself: few V-cycles per solve, r.r < tol, div(u) = 0 to the solver tolerance.

WL_FULLSIZE selects cases by id (comma separated, default: all four)."""
import gc
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = {
    "C2-256^3-f32-sphere": ((256, 256, 256), np.float32, "sphere"),
    "C3-512^3-f32-sphere": ((512, 512, 512), np.float32, "sphere"),
    "C4-1024x1024x512-f32-sphere": ((1024, 1024, 512), np.float32, "sphere"),
    "C5-512^3-f64-torus": ((512, 512, 512), np.float64, "torus"),
}
_want = [w.strip() for w in os.environ.get("WL_FULLSIZE", ",".join(CASES)).split(",")]
IDS = [k for k in CASES if any(k.startswith(w) or w in k for w in _want)]


@pytest.fixture(scope="module")
def S():
    from waterlily_amd import sim
    return sim


@pytest.fixture(scope="module", params=IDS)
def case(request):
    yield CASES[request.param]
    gc.collect()
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def flow(S, case):
    dims, T, _ = case
    U = (2 / 3, -1 / 3, 0.25)
    a = S.Flow(dims, U, T=T, ulam=lambda i, x: U[i])
    yield a, S.MultiLevelPoisson(a.p, a.mu0, a.sigma), U
    del a
    gc.collect()
    torch.cuda.empty_cache()


def rand_like(a, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.rand(a.shape, generator=g, device="cuda", dtype=a.dtype) - 0.5


def test_bc_idempotent_and_uniform_convdiff_zero(S, flow):
    a, ml, U = flow
    S.BC(a.u, U)                                             # (the constructor's exitBC! (Flow.jl:115) leaves the exit plane shifted
    u1 = S.copy_of(a.u)                                      #  by the rounding of its mean-flux correction: BC! resets it once)
    S.BC(a.u, U)
    assert torch.equal(a.u, u1)
    del u1
    S.conv_diff(a.f, a.u, nu=0.01)
    # (the two x-planes next to the exit are excluded: the Flow constructor's exitBC! (Flow.jl:115) shifts the exit
    #  plane by the rounding of its mean-flux correction)
    inner = (slice(1, -3),) + (slice(1, -1),) * 2
    assert float(a.f[inner].abs().max()) == 0.0


def test_operator_symmetry_and_linearity(S, flow, case):
    a, ml, U = flow
    T = case[1]
    lv = ml.levels[0]
    x, y = lv.layout.alloc((), "cuda:0"), lv.layout.alloc((), "cuda:0")
    inner = (slice(1, -1),) * 3
    x[inner] = rand_like(x[inner], 1)
    y[inner] = rand_like(y[inner], 2)
    Ay = S.copy_of(S.mult(ml, y))
    xAy = S.dot(x, Ay)
    del Ay
    Ax = S.copy_of(S.mult(ml, x))
    yAx = S.dot(y, Ax)
    assert abs(xAy - yAx) <= (1e-5 if T == np.float32 else 1e-12) * max(abs(xAy), 1.0)
    x4 = y                                                   # (reuse the buffer: C4 arrays are 2 GB each)
    x4[inner] = x[inner] * 4
    A4x = S.mult(ml, x4)
    assert torch.equal(A4x[inner], Ax[inner] * 4)


def test_restrict_prolongate_identities(S, flow, case):
    a, ml, U = flow
    T = case[1]
    f, c = ml.levels[0], ml.levels[1]
    ci = (slice(1, -1),) * 3
    c.x.zero_()
    c.x[ci] = rand_like(c.x[ci], 3)
    S.prolongate(f.eps, c.x)
    S.restrict(c.r, f.eps)
    # sum of the 8 identical children: 8c up to the roundings of the partial sums 3c,5c,6c,7c
    assert float((c.r[ci] - 8 * c.x[ci]).abs().max()) <= 4 * np.finfo(T).eps * 8 * float(c.x[ci].abs().max())
    f.r.zero_()
    f.r[inside3(f.r)] = rand_like(f.r[inside3(f.r)], 4)
    S.restrict(c.r, f.r)
    sf, sc = float(f.r.double().sum()), float(c.r.double().sum())
    assert abs(sf - sc) <= 1e-6 * max(1.0, float(f.r.double().abs().sum()))


def inside3(a):
    return (slice(1, -1),) * 3


def test_projection_divergence_free_and_uniform_stream(S, flow, case):
    a, ml, U = flow
    dims = case[0]
    S.mom_step(a, ml)
    assert all(1 <= n <= 32 for n in ml.n[-2:])
    assert S.L2p(ml) < 1e-4                                   # solver! tolerance (MultiLevelPoisson.jl:87,95)
    z = S.like(a.p)
    S.divergence(z, a.u)
    assert S.L2(z) < 1e-3
    del z
    ncell = float(np.prod(dims))
    for i in range(3):                                        # impulsive uniform stream stays uniform
        d = (a.u[..., i] - U[i])[(slice(1, -1),) * 3]
        assert float((d.double() ** 2).sum()) < 2e-5 * ncell / 16 ** 3


def test_rows_per_thread_same_bits(S, flow):
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


def _bench_case(dims, T, kind):
    import bench
    return (bench.sphere if kind == "sphere" else bench.donut)(tuple(dims), T)


def test_traffic_saving_switches_do_not_change_a_bit(S, case):
    """The kernels that move fewer bytes than the dense algorithm -- row constants instead of L/iD in coefficient-
    uniform rows (option 9), x += alpha*eps deferred to the direction kernel (8), z' = r*iD recomputed instead of
    stored (13), z = A*eps formed a second time by the update kernel instead of stored (19), body-free rows in BDIM! (3), the chained x/=dt ; x*=dt' pass (14), the shared-flux conv_diff! kernel (18), div(u)
    formed inside residual! (22), the x planes of BC! written by the producing kernel (23), consecutive kernels sweeping in
    opposite directions (30) -- evaluate the same expressions: three steps of the case give
    bit-identical u and p with all of them off."""
    dims, T, kind = case
    if int(np.prod(dims)) > 512 ** 3:
        pytest.skip("C4: the same kernels at four times the cells (55 s); C2, C3 and C5 run this comparison")
    # Float64: options 8, 13 and 19 decide which kernels accumulate pcg!'s dot products and how their grids are cut (the
    # in-kernel finalisation needs 8 and 13; 19 moves r.z' into a 7-point kernel): the SAME terms are summed in a different
    # grouping.  Rounded to Float32 the sums are the same numbers; in Float64 their last bits differ, and with them
    # everything downstream (tools/whichswitch.py) -- so there these three are compared on their own, to rounding, and the
    # other switches (none of which regroups a sum) bit for bit.
    f64 = np.dtype(T) == np.float64
    keys = (3, 9, 14, 18, 22, 23, 30) + (() if f64 else (8, 13, 19))
    regroup = (8, 13, 19)

    def run(off):
        for key in off:
            S.set_option(key, 0)
        if 19 not in off and not f64:
            S.set_option(19, 2)                       # (2 = on every level, also the finest one)
        try:
            sim = _bench_case(dims, T, kind)
            for _ in range(3):
                S.sim_step(sim, remeasure=False)
        finally:
            for key in keys + regroup:
                S.set_option(key, 1)
        nu, nr = S.uniform_rows(sim.pois, 0)
        out = (sim.pois.n[:], list(sim.flow.dt), S.copy_of(sim.flow.u), S.copy_of(sim.flow.p), nu, nr)
        del sim                                       # one simulation at a time (C4: 68 GB each)
        gc.collect()
        torch.cuda.empty_cache()
        return out
    a, b = run(()), run(keys)
    assert a[4] > 0.7 * a[5]                  # most rows of the case are coefficient-uniform (sphere: 94 %, torus: 79 %)
    assert a[0] == b[0] and a[1] == b[1]
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    if f64:
        c = run(regroup)
        assert a[0] == c[0] and np.allclose(a[1], c[1], rtol=1e-12, atol=0)
        assert float((a[2] - c[2]).abs().max()) <= 1e-11 * float(a[2].abs().max())
        assert float((a[3] - c[3]).abs().max()) <= 1e-10 * float(a[3].abs().max())


def test_hydrostatic_force_and_two_steps_of_the_configuration(S, case):
    """The BASELINE configuration itself (bench.py's set-up), two `sim_step!`s from the impulsive start: every solve
    converges in a few V-cycles (the reference's own multigrid bound is n <= 3 on its manufactured problems,
    maintests.jl:112-115; an impulsive start with a body takes one or two more on the first solve), leaves r.r below the
    solver tolerance and a velocity field that is divergence-free to it; forces, dt and u stay finite."""
    dims, T, kind = case
    sim = _bench_case(dims, T, kind)
    # maintests.jl:341-346 in 3-D at full size first, through the whole product path (measure! on the device, the |d| <= 1 band
    # rebuilt from its band cells, wl_pforce): p = y  =>  force = displaced volume * e_y
    m = min(dims)
    vol = 4 / 3 * math.pi * (m / 8) ** 3 if kind == "sphere" else 2 * math.pi ** 2 * (m / 4) * (m / 16) ** 2
    yy = torch.arange(dims[1] + 2, device="cuda", dtype=sim.flow.p.dtype) - 0.5
    sim.flow.p.copy_(yy[None, :, None].expand(*(n + 2 for n in dims)))
    force = S.pressure_force(sim)
    assert np.sum(np.abs(force / vol - np.array([0, 1, 0]))) < 2e-3, force / vol
    sim.flow.p.zero_()
    S.sim_step(sim, remeasure=False)
    f1 = S.pressure_force(sim)
    S.sim_step(sim, remeasure=False)
    assert len(sim.pois.n) == 4 and all(1 <= n <= 5 for n in sim.pois.n), sim.pois.n
    assert sim.pois.n[-1] <= 3
    assert S.L2p(sim.pois) < 1e-4
    z = S.like(sim.flow.p)
    S.divergence(z, sim.flow.u)
    assert S.L2(z) < 1e-3
    assert np.all(np.isfinite(f1)) and np.all(np.isfinite(S.pressure_force(sim)))
    assert 0.0 < sim.flow.dt[-1] <= 10.0
    assert bool(torch.isfinite(sim.flow.u).all())


def test_vtk_snapshot_every_step_at_512_costs_the_stepper_little(tmp_path):
    """SURVEY 8f row 3 at a BASELINE size (C3, 512^3 Float32): a snapshot of u and p (2.2 GB) EVERY step for a burst of steps.
    The stepping thread pays for the device-side pack only (the ring of staging slots lives in HBM); the D2H copy runs on a side
    stream and the file is written by the worker thread: the step time rises by < 10 %, the host side holds ONE pinned buffer
    of a snapshot (resident set + pinned memory < 2 x one velocity field; round 3's writer held the numpy copy, a padded copy, its
    base64 string and an XML tree: > 6 x), and what lands on disk is the field of its step."""
    import resource
    import time
    from waterlily_amd import sim as S
    from waterlily_amd import vtk
    import bench
    sim = bench.sphere((512, 512, 512), np.float32)
    for _ in range(12):                     # past the start-up steps (2-3 V-cycles per solve): both timings in the steady state
        S.sim_step(sim, remeasure=False)

    def steps(n, each=None):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            S.sim_step(sim, remeasure=False)
            if each:
                each()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n
    base = steps(6)
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss * 1024
    out = os.environ.get("WL_VTK_DIR", str(tmp_path))
    wr = vtk.vtkWriter(os.path.join(out, "c3"), dir=os.path.join(out, "C3_DIR"), ring=6, host_buffers=1)
    vtk.write(wr, sim)                      # the first snapshot sizes the staging ring and the pinned buffer (allocations synchronise)
    vtk.flush(wr)
    enq0 = wr.stats["enqueue_s"]
    with_snap = steps(6, lambda: vtk.write(wr, sim))
    p_last = S.to_host(sim.flow.p)
    t0 = time.perf_counter()
    vtk.close(wr)
    drain = time.perf_counter() - t0
    rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss * 1024
    one_field = 3 * 514 ** 3 * 4
    print(f"\n512^3 snapshot every step: {base * 1e3:.2f} -> {with_snap * 1e3:.2f} ms per step (+{(with_snap / base - 1) * 100:.1f} %), "
          f"enqueue {(wr.stats['enqueue_s'] - enq0) / 6 * 1e3:.2f} ms per snapshot, D2H {wr.stats['bytes'] / max(wr.stats['d2h_s'], 1e-9) / 1e9:.1f} GB/s, "
          f"file write {wr.stats['bytes'] / max(wr.stats['write_s'], 1e-9) / 1e9:.2f} GB/s, drain after the burst {drain:.1f} s, "
          f"host memory +{(rss1 - rss0) / one_field:.2f} velocity fields")
    assert with_snap <= 1.10 * base
    assert rss1 - rss0 < 2 * one_field
    assert wr.stats["snapshots"] == 7
    items = vtk.read_pvd(os.path.join(out, "c3.pvd"))
    assert np.array_equal(np.asarray(vtk.read_vti(items[-1][1])["Pressure"]), p_last)
    for _, path in items:
        os.remove(path)
