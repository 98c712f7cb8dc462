#!/usr/bin/env python3
"""bench.py -- MLUPS (cell-updates/s) per `sim_step!` of the 3-D sphere case (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W

One "step" = one `sim_step!(sim; remeasure=false)` = one `mom_step!` (predictor + corrector, both pressure
solves, CFL).  Workload at N=1: BASELINE.json configs[2], the configuration the metric is quoted on:
3-D sphere, 512^3, Float32, Re=3700, uniform inflow (geometry as README.md:118-125 of the reference).
Inputs are synthetic and resident in HBM before the timed region.  Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the GPU box gives one GPU a 16-core CPU share (os.cpu_count() reports the whole host): bound OpenMP to it
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))
os.environ.setdefault("OMP_PROC_BIND", "close")

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s float4 copy)

# ALGORITHMIC bytes per processed cell, in units of the element size T (each distinct array element the kernel's
# operator reads once + writes once, dense -- SURVEY.md 8(d), DESIGN.md "kernels and their algorithmic bytes").
# Keyed by libwlhip kernel class: (dense, never_moved, skipped_in_uniform_rows).  The kernels move LESS than `dense`:
# `never_moved` is the diagonal D, recomputed from the six face coefficients; `skipped_in_uniform_rows` are the
# coefficient arrays (L x3, iD) that are not loaded in x-rows whose coefficients are one number (rows clear of the
# body: 94 % of the rows of the 512^3 sphere case).  roofline.achieved uses `dense` (the contract's figure, which can
# therefore exceed the HBM peak); roofline.moved uses dense - never - phi*skipped, the bytes the kernel has to move.
ALG_T = {
    "conv_diff": (10.5, 0, 0),         # fused conv_diff!+BDIM#1: u(3) [+u0(3) corrector] + V(3) -> f(3) [+u0(3) predictor]
    "bdim": (22.5, 0, 0),              # BDIM#2: f(3) V(3) mu0(3) mu1(9) [+u(3) corrector] -> u(3): 21T/24T (dense)
    "pcg_mult_dot": (6.0, 1, 3),       # eps, L(3), D -> z   (+ z.eps partial)
    "pcg_update": (13.0 / 3, 0, 5.0 / 6),  # r, z, iD -> r (+ r.(r iD) partial): 4T; the 6th iteration x, eps, r, z -> x, r: 6T
    "pcg_direction": (6.0, 0, 1),      # x, eps, r, iD -> x, eps   (x += alpha eps ; eps = beta eps + r iD)
    "pcg_init": (3.0, 0, 1),           # r, iD -> eps
    "smooth": (9.0, 1, 3.5),           # fused Jacobi!+increment! r,iD,x,D,L(3) -> r,x; fused prolongate!+increment! skips L only
    "jacobi": (3.0, 0, 0),
    "increment": (9.0, 1, 3),
    "residual": (8.0, 1, 3),           # x, L(3), D, z, iD -> r
    "restrict": (9.0, 0, 0),
    "prolongate": (2.0, 0, 0),
    "dot": (1.0, 0, 0),
    "div": (4.0, 0, 0),
    "correct": (10.0, 0, 0),           # u(3) rw, L(3), x
    "scale": (2.0, 0, 0),
    "cfl": (4.0, 0, 0),
    "copy": (2.0, 0, 0),
}


def sphere(dims, T, Re=3700.0, device="cuda:0", padded=True):
    """reference README.md:118-125: radius=m/8, center=m/2-1, L=2radius, nu=U*L/Re (m = shortest side; the
    sphere sits at the same x,y position and in the middle of the z extent)"""
    import torch
    from waterlily_amd import sim as S
    from waterlily_amd.body import AutoBody, norm2
    m = min(dims)
    radius = m / 8
    cx, cy, cz = m / 2 - 1, dims[1] / 2 - 1, dims[2] / 2 - 1
    body = AutoBody(lambda x, t: torch.sqrt((x[0] - cx) ** 2 + (x[1] - cy) ** 2 + (x[2] - cz) ** 2) - radius)
    return S.Simulation(tuple(dims), (1.0, 0.0, 0.0), 2 * radius, nu=2 * radius / Re, body=body, T=T, device=device,
                        padded=padded)


def donut(dims, T, Re=1000.0, device="cuda:0", padded=True):
    """BASELINE configs[4]: torus sdf(x) = |(x1-c, |(x2-c, x3-c)| - R)| - r, c = m/2, R = m/4, r = m/16,
    L = R, Re = 1000 (the reference only links its donut example, README.md:53)."""
    import torch
    from waterlily_amd import sim as S
    from waterlily_amd.body import AutoBody
    m = min(dims)
    c, R, r = m / 2, m / 4, m / 16

    def sdf(x, t):
        ring = torch.sqrt((x[1] - c) ** 2 + (x[2] - dims[2] / 2) ** 2) - R
        return torch.sqrt((x[0] - c) ** 2 + ring ** 2) - r
    return S.Simulation(tuple(dims), (1.0, 0.0, 0.0), R, nu=R / Re, body=AutoBody(sdf), T=T, device=device, padded=padded)


def cpu_baseline(size: int, steps: int):
    """The CPU restatement of the reference (oracle/, OpenMP) timed on the host cores on a bounded sample."""
    from oracle import wl_oracle as O
    from waterlily_amd import body as B
    from waterlily_amd.body import AutoBody, norm2
    m = size
    radius, center = m / 8, m / 2 - 1
    body = AutoBody(lambda x, t: norm2(x - center) - radius)
    s = O.Simulation((m, m, m), (1.0, 0.0, 0.0), 2 * radius, nu=2 * radius / 3700.0, body=body, T=np.float32,
                     measure_fn=B.measure_fields, nds_fn=B.nds_band)
    O.sim_step(s, remeasure=False)  # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        O.sim_step(s, remeasure=False)
    dt = time.perf_counter() - t0
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    return {"value": m ** 3 * steps / dt / 1e6, "unit": "MLUPS", "cores": cores, "kind": "port",
            "sample": f"{m}^3 sphere Re=3700 f32, {steps} steps after 1 warm-up, remeasure=false, "
                      f"V-cycles/step={s.pois.n[-2:]}"}


def class_table(L):
    names = {}
    k = 0
    while True:
        nm = L.wl_kernel_name(k).decode()
        if nm == "?":
            break
        names[nm] = k
        k += 1
    return names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=512, help="cells per side (BASELINE config: 512)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--cpu-size", type=int, default=192)
    ap.add_argument("--cpu-steps", type=int, default=30, help="timed steps of the CPU baseline (192^3: about 10 s on 16 cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--body", default="sphere", choices=["sphere", "donut"], help="donut: BASELINE configs[4] (use with --dtype f64)")
    ap.add_argument("--layout", default="padded", choices=["padded", "dense"],
                    help="padded: rows 128-B aligned (default); dense: the reference's column-major layout (pitch N+2)")
    ap.add_argument("--kernel", default=None, help="force the kernel class reported in `roofline`")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "host"],
                    help="multi-rank transport: rccl (one GPU per rank) or host (gloo staging; lets ranks share a GPU, tests)")
    ap.add_argument("--grid", type=int, nargs=3, default=None,
                    help="explicit GLOBAL grid nx ny nz (strong scaling, e.g. 1024 1024 512 = BASELINE configs[3])")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")

    from waterlily_amd import _lib
    from waterlily_amd import dist as wd
    from waterlily_amd import sim as S
    L = _lib.lib()
    T = np.float32 if args.dtype == "f32" else np.float64
    tsz = np.dtype(T).itemsize
    transport = "rccl" if args.comm == "rccl" else "host staging over gloo"
    if args.comm == "host":
        local = local % max(1, torch.cuda.device_count())
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)
    if world > 1 and args.comm == "rccl":
        # one process per GPU; the z axis is cut into `world` slabs, halos + scalar all-reduces run over RCCL (xGMI)
        dist.init_process_group("nccl", device_id=torch.device(dev))
        ok = 1
        try:
            wd.init_rccl()
        except Exception as e:   # e.g. ncclCommInitRank refused: fall back to host staging over gloo rather than no number
            print(f"[bench] rank {rank}: RCCL communicator failed ({e}); falling back to --comm host", file=sys.stderr, flush=True)
            ok = 0
        flag = torch.tensor([ok], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if ok:
                wd.finalize()
            wd.init_host(dist.new_group(backend="gloo"))
            transport = "host staging over gloo (RCCL fallback)"
    elif world > 1:
        dist.init_process_group("gloo")
        wd.init_host()
    m = args.size
    # N=1: the BASELINE 512^3 cube.  N>1 (default): WEAK scaling -- every GPU keeps a 512x512x512 slab, i.e. the
    # global grid is 512 x 512 x 512N; --grid gives an explicit global grid instead (strong scaling).
    dims = tuple(args.grid) if args.grid else (m, m, m * world)
    scaling = "strong" if args.grid else "weak"
    sim = (sphere if args.body == "sphere" else donut)(dims, T, device=dev, padded=(args.layout == "padded"))
    ncell_global = int(np.prod(dims))
    ncell = ncell_global // world            # cells per rank: threshold for "finest level" launches
    names = class_table(L)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # warm-up; the last warm-up step times EVERY finest-level launch with hipEvents to find the dominant kernel
    per_class = {}
    for w in range(args.warmup):
        S.sim_step(sim, remeasure=False)
    sync()
    # each heavy finest-level class is timed with hipEvents in one extra warm-up step to find the dominant kernel
    heavy = ["pcg_mult_dot", "pcg_update", "pcg_direction", "smooth", "conv_diff", "bdim", "residual"]
    if args.kernel:
        dominant = args.kernel
    else:
        for nm in heavy:
            _lib.check(L.wl_prof_reset())
            _lib.check(L.wl_prof_select(names[nm], int(0.9 * ncell)))
            S.sim_step(sim, remeasure=False)
            nl, nc, ms = C.c_int64(), C.c_int64(), C.c_double()
            _lib.check(L.wl_prof_timed(C.byref(nl), C.byref(nc), C.byref(ms)))
            per_class[nm] = {"launches": nl.value, "ms": ms.value}
        dominant = max(per_class, key=lambda k: per_class[k]["ms"])

    # timed region: only the dominant class is bracketed by hipEvents (on the library's stream)
    _lib.check(L.wl_prof_reset())
    _lib.check(L.wl_prof_select(names[dominant], int(0.9 * ncell)))
    n0 = len(sim.pois.n)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        S.sim_step(sim, remeasure=False)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:  # MAX over ranks
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.comm == "rccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    nl, nc, ms = C.c_int64(), C.c_int64(), C.c_double()
    _lib.check(L.wl_prof_timed(C.byref(nl), C.byref(nc), C.byref(ms)))
    _lib.check(L.wl_prof_select(-1, 0))
    vcycles = sim.pois.n[n0:]

    mlups = ncell_global * args.steps / elapsed / 1e6
    avg_ms = ms.value / max(1, nl.value)
    dense, never, skipped = ALG_T[dominant]
    n_uni, n_rows = S.uniform_rows(sim.pois, 0)
    phi = n_uni / max(1, n_rows)                       # share of x-rows whose L / iD loads are skipped
    cells_per_launch = nc.value / max(1, nl.value)
    alg_bytes = dense * tsz * cells_per_launch
    moved_bytes = (dense - never - phi * skipped) * tsz * cells_per_launch
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    moved_rate = moved_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile):
        try:
            traffic = json.load(open(tfile)).get(f"{dominant}@{m}^3/{args.dtype}") if not args.grid else None
        except Exception:
            traffic = None
    # the north-star's named kernel: the V-cycle smoother (fused Jacobi!+increment! and prolongate!+increment!), from
    # the per-class hipEvent pass above (finest-level launches only)
    smoother = None
    if per_class.get("smooth", {}).get("launches"):
        sm = per_class["smooth"]
        sm_ms = sm["ms"] / sm["launches"]
        d_, n_, k_ = ALG_T["smooth"]
        sm_alg = d_ * tsz * ncell
        sm_mov = (d_ - n_ - phi * k_) * tsz * ncell
        smoother = {"kernel": "smooth", "avg_launch_ms": sm_ms, "launches": sm["launches"],
                    "achieved": sm_alg / (sm_ms * 1e-3) / 1e9, "frac": sm_alg / (sm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "moved_GB/s": sm_mov / (sm_ms * 1e-3) / 1e9, "moved_frac": sm_mov / (sm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "unit": "GB/s", "algorithmic_bytes_per_launch": sm_alg}
    out = {
        "metric": "MLUPS (cell-updates/s) per sim_step!, 3D sphere", "value": mlups, "unit": "MLUPS",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"3D {args.body} {dims[0]}x{dims[1]}x{dims[2]}, Re={3700 if args.body == 'sphere' else 1000}, {args.dtype}, uniform inflow, "
                               f"remeasure=false" + (", dense layout" if args.layout == "dense" else "") + (" (BASELINE configs[2])" if world == 1 and not args.grid and m == 512
                                                     and args.dtype == "f32" and args.body == "sphere" else "" if world == 1 else
                                                     f", z-slabs over {world} GPUs ({transport})"),
                   "vcycles_per_solve": vcycles[:6], "mean_vcycles_per_step": float(np.sum(vcycles)) / args.steps},
        "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "launches": nl.value,
                     "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": alg_bytes,
                     "moved": {"bytes_per_launch": moved_bytes, "GB/s": moved_rate, "frac": moved_rate / HBM_PEAK_GBS,
                               "uniform_row_fraction": phi,
                               "note": "achieved/frac use the dense algorithmic bytes of SURVEY 8(d); the kernel skips the "
                                       "loads of L/iD in coefficient-uniform rows and recomputes D, so it moves only "
                                       "`moved.bytes_per_launch` (compare `traffic`, the PMC-measured bytes)"},
                     "smoother": smoother, "per_class_ms_one_step": per_class},
    }
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args.cpu_size, args.cpu_steps)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        wd.finalize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
