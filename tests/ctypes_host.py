#!/usr/bin/env python3
"""A torch-free host of libwlhip.so: the call sequence of the reference-side binding, exercised on a GPU.

This is what `julia/WaterLilyHIPNativeExt.jl` does (lines cited per step) written with nothing but ctypes + numpy: the
library has to own its HIP context, its allocations and its stream -- no torch creates the context, hands over memory or
sets the current device.  (Every other GPU test lets torch do those three things.)

    HIPArray(zeros(T, Nd))           -> wl_malloc + wl_memset0 + wl_h2d_2d  (shim `HIPArray`; src/Flow.jl:114-118 `zeros(T,Nd) |> f`)
    Flow / MultiLevelPoisson         -> wl_flow_create, wl_mg_create  (shim `handle`; strides from the array's pitch)
    measure!(flow, body) epilogue    -> wl_flow_update, wl_mg_update  (src/Body.jl:31-53, src/MultiLevelPoisson.jl:62-68)
    mom_step!(flow, pois)            -> wl_mom_step                   (shim :260-280; src/Flow.jl:153-169)
    pressure_force(sim)              -> wl_pforce                     (src/Metrics.jl:94-100)
    Array(flow.u), Array(flow.p)     -> wl_d2h ; finalizers           -> wl_free, wl_*_destroy

Inputs and expected outputs: tests/golden/sim_*.npz (coefficient fields measured by the oracle's geometry, initial velocity,
u / p / pois.n / dt / force after `nsteps` steps).  The body term of pressure_force (n̂·kern, Metrics.jl:84-87: a user
closure in the reference, evaluated on the host side of the ABI) comes from oracle/geometry.py -- test infrastructure.

Two layouts: "pitched" (default: what the shim's HIPArray allocates -- row stride rounded up to 128 bytes, first interior element
of every row on a 128-byte boundary, host copies by wl_h2d_2d / wl_d2h_2d) and "dense" (the reference's own strides, pitch N+2,
plain wl_h2d / wl_d2h).

    python tests/ctypes_host.py sim_3d_f32 [pitched|dense] [libwlhip.so]

Prints one JSON line; exit code 0 = every comparison passed.  `import torch` never happens (asserted at the end)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Grid(C.Structure):   # wl_grid, include/wlhip.h
    _fields_ = [("D", C.c_int32), ("n", C.c_int32 * 3), ("s", C.c_int64 * 3), ("sc", C.c_int64), ("nzg", C.c_int32),
                ("kz0", C.c_int32), ("own_lo", C.c_int32), ("own_hi", C.c_int32), ("zring", C.c_int32)]


class Level(C.Structure):  # wl_level_desc
    _fields_ = [("g", Grid)] + [(k, C.c_void_p) for k in ("L", "D", "iD", "x", "eps", "r", "z")]


class FlowDesc(C.Structure):  # wl_flow_desc
    _fields_ = [("g", Grid)] + [(k, C.c_void_p) for k in ("u", "u0", "f", "p", "sigma", "V", "mu0", "mu1")] + [
        ("nu", C.c_double), ("exitBC", C.c_int32), ("perdir_mask", C.c_int32)]


def main(name, lib_path, layout="pitched"):
    L = C.CDLL(lib_path)
    L.wl_last_error.restype = C.c_char_p

    def chk(rc, what):
        if rc != 0:
            raise SystemExit(f"{what} failed ({rc}): {L.wl_last_error().decode(errors='replace')}")

    ndev = C.c_int()
    chk(L.wl_device_count(C.byref(ndev)), "wl_device_count")
    if ndev.value < 1:
        raise SystemExit("no GPU")
    chk(L.wl_set_device(0), "wl_set_device")

    g = dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")))
    dims = tuple(int(d) for d in g["dims"])
    D = len(dims)
    T = g["u"].dtype
    wlt = 0 if T == np.float32 else 1
    owned = []

    al = 128 // T.itemsize
    for fn in ("wl_h2d_2d", "wl_d2h_2d"):
        getattr(L, fn).argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t]

    def pitch_of(n0):     # elements between consecutive x-rows
        return n0 if layout == "dense" else -(-n0 // al) * al

    def field_grid(Ng):   # wl_grid of a field of spatial extents Ng (components: one block of rows each)
        gr = Grid()
        n = tuple(Ng) + (1,) * (3 - D)
        pt = pitch_of(n[0])
        gr.D = D
        gr.n[:] = n
        gr.s[:] = (1, pt, pt * n[1])
        gr.sc = pt * n[1] * n[2]
        return gr

    def alloc(shape):     # HIPArray{T,N}(dims): zeroed; returns the pointer to element [1,1,...]
        rows = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        lead = 0 if layout == "dense" else al - 1
        nb = (lead + pitch_of(shape[0]) * rows + al) * T.itemsize
        base = C.c_void_p()
        chk(L.wl_malloc(C.byref(base), C.c_size_t(nb)), "wl_malloc")
        chk(L.wl_memset0(base, C.c_size_t(nb)), "wl_memset0")
        owned.append(base)
        p = C.c_void_p(base.value + lead * T.itemsize)
        assert layout == "dense" or (p.value + T.itemsize) % 128 == 0       # the first interior element of a row is aligned
        return p

    def device(h):        # HIPArray(h::Array): copyto!(HIPArray(size(h)), h)
        h = np.asfortranarray(h, dtype=T)
        p = alloc(h.shape)
        rows = int(np.prod(h.shape[1:]))
        w = h.shape[0] * T.itemsize
        if layout == "dense":
            chk(L.wl_h2d(p, h.ctypes.data_as(C.c_void_p), C.c_size_t(h.nbytes)), "wl_h2d")
        else:
            chk(L.wl_h2d_2d(p, pitch_of(h.shape[0]) * T.itemsize, h.ctypes.data_as(C.c_void_p), w, w, rows), "wl_h2d_2d")
        return p

    def table(h):         # a plain device buffer (band indices, n*kern vectors): no rows, no pitch
        h = np.ascontiguousarray(h)
        p = C.c_void_p()
        chk(L.wl_malloc(C.byref(p), C.c_size_t(max(1, h.nbytes))), "wl_malloc")
        chk(L.wl_h2d(p, h.ctypes.data_as(C.c_void_p), C.c_size_t(h.nbytes)), "wl_h2d")
        owned.append(p)
        return p

    zeros = alloc          # fill!(similar(x), 0)

    def host(p, shape):   # Array(a::HIPArray)
        h = np.empty(shape, dtype=T, order="F")
        rows = int(np.prod(shape[1:]))
        w = shape[0] * T.itemsize
        if layout == "dense":
            chk(L.wl_d2h(h.ctypes.data_as(C.c_void_p), p, C.c_size_t(h.nbytes)), "wl_d2h")
        else:
            chk(L.wl_d2h_2d(h.ctypes.data_as(C.c_void_p), w, p, pitch_of(shape[0]) * T.itemsize, w, rows), "wl_d2h_2d")
        return h

    Ng = tuple(n + 2 for n in dims)
    # Flow (src/Flow.jl:112-121): u from the fixture (uλ applied, BC!, exitBC! already done by the oracle's constructor)
    u, u0 = device(g["u_init"]), device(g["u_init"])
    f, p, sigma = zeros(Ng + (D,)), zeros(Ng), zeros(Ng)
    V, mu0, mu1 = device(g["V"]), device(g["mu0"]), device(g["mu1"])
    fd = FlowDesc()
    fd.g = field_grid(Ng)
    for k, v in (("u", u), ("u0", u0), ("f", f), ("p", p), ("sigma", sigma), ("V", V), ("mu0", mu0), ("mu1", mu1)):
        setattr(fd, k, v)
    m = dims[-1]
    R = m / 8
    fd.nu, fd.exitBC, fd.perdir_mask = 2 * R / float(g["Re"]), 0, 0
    flow = C.c_void_p()
    chk(L.wl_flow_create(C.byref(flow), wlt, C.byref(fd)), "wl_flow_create")

    # MultiLevelPoisson(p, μ₀, σ) (src/MultiLevelPoisson.jl:18-25,36-37,51-59): level shapes on the host, arrays from wl_malloc
    shapes = [Ng]
    while all(n % 2 == 0 and n > 4 for n in shapes[-1]) and len(shapes) <= 10:
        shapes.append(tuple(1 + n // 2 for n in shapes[-1]))
    levels = (Level * len(shapes))()
    for l, sh in enumerate(shapes):
        levels[l].g = field_grid(sh)
        if l == 0:
            levels[l].x, levels[l].L, levels[l].z = p, mu0, sigma          # aliasing, src/WaterLily.jl:77
        else:
            levels[l].x, levels[l].L, levels[l].z = zeros(sh), zeros(sh + (D,)), zeros(sh)
        for k in ("D", "iD", "eps", "r"):
            setattr(levels[l], k, zeros(sh))
    mg = C.c_void_p()
    chk(L.wl_mg_create(C.byref(mg), wlt, len(shapes), levels, 0), "wl_mg_create")
    chk(L.wl_flow_update(flow), "wl_flow_update")          # end of measure!(flow, body)
    chk(L.wl_mg_update(mg), "wl_mg_update")                # update!(pois)

    # sim_step!(sim; remeasure=false) x nsteps (src/WaterLily.jl:98-109 -> mom_step!)
    L.wl_mom_step.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.POINTER(C.c_double), C.c_void_p, C.c_void_p,
                              C.POINTER(C.c_double), C.POINTER(C.c_int)]
    U = (C.c_double * 3)(1.0, 0.0, 0.0)
    dts, ns = [float(T.type(0.25))], []
    for _ in range(int(g["nsteps"])):
        dtn, n2 = C.c_double(), (C.c_int * 2)()
        chk(L.wl_mom_step(flow, mg, dts[-1], U, None, None, C.byref(dtn), n2), "wl_mom_step")
        ns += [n2[0], n2[1]]
        dts.append(dtn.value)

    # pressure_force: the body term from the checker's geometry, the integral in the library
    from oracle import geometry as G
    idx, nds = G.nds_band(G.Body(G.Sphere(m / 2 - 1, R)), dims, t=0.0)
    sub = np.unravel_index(np.asarray(idx, dtype=np.int64), Ng, order="F")       # dense cell index -> element offset in the field
    strides = (1, pitch_of(Ng[0]), pitch_of(Ng[0]) * Ng[1])
    idx = np.ascontiguousarray(sum(q.astype(np.int64) * st for q, st in zip(sub, strides)), dtype=np.int64)
    nds = np.ascontiguousarray(nds, dtype=np.float64)
    L.wl_pforce.argtypes = [C.c_int, C.POINTER(Grid), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_double)]
    force = (C.c_double * 3)()
    gr = field_grid(Ng)
    chk(L.wl_pforce(wlt, C.byref(gr), p, table(idx), table(nds), len(idx), force), "wl_pforce")   # nds[b*D + c]

    uh, ph = host(u, Ng + (D,)), host(p, Ng)
    chk(L.wl_mg_destroy(mg), "wl_mg_destroy")
    chk(L.wl_flow_destroy(flow), "wl_flow_destroy")
    for q in owned:
        chk(L.wl_free(q), "wl_free")

    f32 = T == np.float32
    tol = 5e-4 if f32 else 1e-10          # the tolerances of tests/test_golden.py::test_hip_sim_golden
    res = {
        "case": name, "layout": layout, "n": ns, "n_expected": [int(v) for v in g["n"]],
        "du": float(np.max(np.abs(uh - g["u"])) / np.max(np.abs(g["u"]))),
        "dp": float(np.max(np.abs(ph - g["p"])) / np.max(np.abs(g["p"]))),
        "ddt": float(np.max(np.abs(np.array(dts) - g["dt"]) / g["dt"])),
        "force": [force[c] for c in range(D)], "force_expected": [float(v) for v in g["force"]],
        "torch_imported": "torch" in sys.modules, "tol": tol,
    }
    ok = (res["n"] == res["n_expected"] and res["du"] <= tol and res["dp"] <= 10 * tol and res["ddt"] <= tol
          and np.allclose(res["force"], res["force_expected"], rtol=1e-3 if f32 else 1e-8, atol=1e-5 if f32 else 1e-10)
          and not res["torch_imported"])
    res["ok"] = bool(ok)
    print(json.dumps(res))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "waterlily_amd", "libwlhip.so"),
                  sys.argv[2] if len(sys.argv) > 2 else "pitched"))
