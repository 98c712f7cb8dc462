#!/usr/bin/env python3
"""bench.py -- MLUPS (cell-updates/s) per `sim_step!` of the 3-D sphere case (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one `sim_step!(sim; remeasure=false)` = one `mom_step!` (predictor + corrector, both pressure
solves, CFL).  Workload at N=1: BASELINE.json configs[2] (C3), the configuration the metric is quoted on:
3-D sphere, 512^3, Float32, Re=3700, uniform inflow (geometry as README.md:118-125 of the reference).
Workload at N>1: BASELINE.json configs[3] (C4), 1024 x 1024 x 512 Float32, cut into N z-slabs (one process per GPU, RCCL):
STRONG scaling -- the north star's ">= 6x at 8 GPUs vs 1 GPU" case; `--weak` / `--size` / `--grid` choose other grids.
Started by a launcher (`python -m torch.distributed.run ... bench.py --gpus N`, WORLD_SIZE set) the process is one rank; started
bare with N > 1 it launches its N ranks itself as children (before importing torch: the parent never touches a GPU), relays
rank 0's line and exits with their status.
Inputs are synthetic and resident in HBM before the timed region.  Prints ONE JSON line (rank 0).

JSON extras:
  roofline      the kernel class that takes most of a step (hipEvents on the library's stream inside the timed region):
                `achieved` = the bytes the kernel HAS TO move per launch (its own operator's compulsory traffic: every
                distinct array element it needs read once + written once; coefficient arrays are not needed in
                coefficient-uniform rows, D is recomputed -- DESIGN.md section 4) / average launch duration, so frac <= 1
                by construction; `dense` = the same with SURVEY 8(d)'s dense per-cell figure of the reference operator
                (secondary: can exceed the peak, the kernel does not move those bytes); `traffic` = HBM bytes per launch
                measured with rocprofv3 PMC counters for the SAME launch mix (profiles/traffic.json, see profiles/parse_pmc.py)
  smoother      the same three figures for the V-cycle smoother Jacobi!+increment! (the north-star's >=40 % kernel) and
  prolong_increment   for the fused prolongate!+increment! kernel, each on its own (finest level only)
  cpu_baseline  the CPU restatement of the reference (oracle/, OpenMP) on BASELINE.md section 3's two CPU points:
                C2 256^3 Float32 sphere (`value`) and C1 2-D circle 192x64 Float64 (`c1`)  (N=1 only)
  layout_dense  N=1: the same case once more in the reference's dense strides (pitch N+2: what a binding that hands over
                Julia's own arrays would give), 5 steps in the steady state -- the headline runs on pitched rows, which is
                what the reference-side binding allocates
  one_gpu, speedup_vs_1gpu   N>1, strong scaling: the same global grid on ONE GPU (rank 0's, after the N-GPU run, same build)
  loopback      --comm loopback: one process plays one rank of an N-way split (measurement tool)
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
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver (before HIP starts)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s float4 copy)

# Bytes per processed cell in units of the element size T, keyed by libwlhip kernel class:
#   dense : SURVEY.md 8(d) -- every distinct array element of the REFERENCE operator read once + written once
#   need  : what the fused kernel has to move in a coefficient-uniform x-row (row constants replace L and iD, D is
#           recomputed from L): its own compulsory traffic
#   extra : the additional arrays it has to load in a row that is NOT uniform (L x3, iD)
# bytes it has to move per cell = need + (1 - phi) * extra, phi = share of coefficient-uniform rows (0.94 for the 512^3 sphere)
ALG_T = {
    #                  dense   need    extra
    # u0.=u + conv_diff! + accelerate! + BDIM! (#1 everywhere, #2 + scale_u! on the body-free rows: the velocity arrays take
    # turns).  predictor: u(3) -> f(3), u'(3); corrector: u'(3), u0(3) -> f(3), u(3): mean of the two
    # launches of a step; V(3) only in the rows that hold a body (priced with the uniform-row share, which is that share)
    "conv_diff":      (34.5,   10.5,   3.0),
    "bdim":           (22.5,   22.5,   0.0),   # BDIM!#2 on the busy rows alone: u(3), f(3), V(3), mu0(3), mu1(9) -> u(3)
    "pcg_mult_dot":   (6.0,    2.0,    3.0),   # eps -> z (+ z.eps); L(3) in non-uniform rows; D recomputed
    "pcg_update":     (13 / 3, 3.5,    5 / 6), # 5 of 6: r,z -> r (+ r.(r iD)); the 6th: x,eps,r,z -> x,r
    "pcg_direction":  (6.0,    5.0,    1.0),   # x,eps,r -> x,eps
    "pcg_init":       (3.0,    2.0,    1.0),   # r -> eps
    "smooth":         (9.0,    4.0,    4.0),   # Jacobi!+increment!: r,x -> r,x ; iD, L(3) in non-uniform rows
    "prolongate":     (9.125,  5.125,  4.0),   # prolongate!+increment! (+ start of pcg!): r,x,(coarse x) -> r,x,eps
    "increment":      (9.0,    5.0,    3.0),
    "residual":       (12.0,   5.0,    4.0),   # project!: z=div(u) formed inside residual! (div 4T + residual! 8T as written): u(3),x -> r
    "correct":        (10.0,   7.0,    3.0),   # u(3) rw, x ; L(3) in non-uniform rows
    "div":            (4.0,    4.0,    0.0),
    "cfl":            (4.0,    4.0,    0.0),
    "scale":          (2.0,    2.0,    0.0),
    "restrict":       (9.0,    9.0,    0.0),
}


def sphere(dims, T, Re=3700.0, device="cuda:0", padded=True, fit=None):
    """reference README.md:118-125: radius=m/8, center=m/2-1, L=2radius, nu=U*L/Re (m = shortest side; the
    sphere sits at the same x,y position and in the middle of the z extent).  fit = (z_lo, z_hi) (loopback runs): the
    sphere is shrunk to fit between those planes and centred there."""
    import torch
    from waterlily_amd import sim as S
    from waterlily_amd.body import AutoBody
    m = min(dims)
    radius = m / 8
    cx, cy, cz = m / 2 - 1, dims[1] / 2 - 1, dims[2] / 2 - 1
    if fit is not None:
        radius, cz = min(radius, 0.375 * (fit[1] - fit[0])), 0.5 * (fit[0] + fit[1])
    body = AutoBody(lambda x, t: torch.sqrt((x[0] - cx) ** 2 + (x[1] - cy) ** 2 + (x[2] - cz) ** 2) - radius)
    return S.Simulation(tuple(dims), (1.0, 0.0, 0.0), 2 * radius, nu=2 * radius / Re, body=body, T=T, device=device,
                        padded=padded)


def donut(dims, T, Re=1000.0, device="cuda:0", padded=True, fit=None):
    """BASELINE configs[4]: torus sdf(x) = |(x1-c, |(x2-c, x3-c)| - R)| - r, c = m/2, R = m/4, r = m/16,
    L = R, Re = 1000 (the reference only links its donut example, README.md:53).  fit: as in sphere()."""
    import torch
    from waterlily_amd import sim as S
    from waterlily_amd.body import AutoBody
    m = min(dims)
    c, R, r = m / 2, m / 4, m / 16
    cz = dims[2] / 2
    if fit is not None:
        R = min(R, 0.3 * (fit[1] - fit[0]))
        r, cz = R / 4, 0.5 * (fit[0] + fit[1])

    def sdf(x, t):
        ring = torch.sqrt((x[1] - c) ** 2 + (x[2] - cz) ** 2) - R
        return torch.sqrt((x[0] - c) ** 2 + ring ** 2) - r
    return S.Simulation(tuple(dims), (1.0, 0.0, 0.0), R, nu=R / Re, body=AutoBody(sdf), T=T, device=device, padded=padded)


def moving_cylinder(dims, T, Re=1000.0, device="cuda:0", padded=True):
    """A circular cylinder (axis along z) translating with unit speed along x through fluid at rest, re-measured EVERY step
    (`sim_step!`'s default, remeasure=true): `norm2(x[1:2] .- center) - radius` under `map(x,t) = x .- (U t, 0, 0)`, the 3-D
    moving-cylinder case of the reference's benchmark suite (README.md:145-151, WaterLily-Benchmarks).  A parametric body:
    measure! and the changed-rows update!(pois) run as HIP kernels (csrc/wl_measure.h)."""
    from waterlily_amd import body as B
    from waterlily_amd import sim as S
    m = min(dims)
    radius = m / 16
    body = B.Cylinder((m / 4, dims[1] / 2, 0.0), radius, 3, map=B.translation(3, v=(1.0, 0.0, 0.0)))
    return S.Simulation(tuple(dims), (0.0, 0.0, 0.0), 2 * radius, U=1.0, nu=2 * radius / Re, body=body, T=T, device=device, padded=padded)


def tgv(dims, T, Re=1600.0, device="cuda:0", padded=True):
    """3-D Taylor-Green vortex in the box (no body): the reference's README / benchmark TGV case, u = (-sin x cos y cos z,
    cos x sin y cos z, 0) with kappa = pi / L, L = m / 2"""
    from waterlily_amd import sim as S
    m = min(dims)
    Lc = m / 2
    k = np.pi / Lc

    def ulam(i, x):
        if i == 0:
            return -np.sin(k * x[0]) * np.cos(k * x[1]) * np.cos(k * x[2])
        if i == 1:
            return np.cos(k * x[0]) * np.sin(k * x[1]) * np.cos(k * x[2])
        return 0.0 * x[0]
    return S.Simulation(tuple(dims), (0.0, 0.0, 0.0), Lc, U=1.0, nu=Lc / Re, ulam=ulam, T=T, device=device, padded=padded)


def _cpu_case(dims, T, Re, steps):
    """one CPU point: the oracle (oracle/wl_oracle.c, OpenMP; body measured by oracle/geometry.py) on the README's
    circle / sphere case, `steps` steps after one warm-up step, remeasure=false"""
    from oracle import geometry as G
    from oracle import wl_oracle as O
    m = dims[-1]
    radius, center = m / 8, m / 2 - 1
    U = (1.0,) + (0.0,) * (len(dims) - 1)
    s = O.Simulation(dims, U, 2 * radius, nu=2 * radius / Re, body=G.Body(G.Sphere(center, radius)), T=T)
    O.sim_step(s, remeasure=False)  # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        O.sim_step(s, remeasure=False)
    dt = time.perf_counter() - t0
    return int(np.prod(dims)) * steps / dt / 1e6, s.pois.n[-2:], dt


def cpu_baseline(size: int, steps: int, c1_steps: int):
    """C2 (3-D sphere 256^3 Float32 Re=3700) and C1 (2-D circle 192x64 Float64 Re=100)."""
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    v2, n2, t2 = _cpu_case((size,) * 3, np.float32, 3700.0, steps)
    v1, n1, t1 = _cpu_case((192, 64), np.float64, 100.0, c1_steps)
    return {"value": v2, "unit": "MLUPS", "cores": cores, "kind": "port",
            "sample": f"C2: {size}^3 sphere Re=3700 f32, {steps} steps after 1 warm-up ({t2:.1f} s), remeasure=false, "
                      f"V-cycles/step={n2}",
            "c1": {"value": v1, "unit": "MLUPS", "cores": cores,
                   "sample": f"C1: 2-D circle 192x64 Re=100 f64, {c1_steps} steps after 1 warm-up ({t1:.2f} s), "
                             f"remeasure=false, V-cycles/step={n1}"}}


def _csrc_digest():
    try:
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        import parse_pmc
        return parse_pmc.csrc_digest()
    except Exception:
        return None


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


C4_GRID = (1024, 1024, 512)   # BASELINE.json configs[3]: the strong-scaling case of the north star


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=None,
                    help="cells per side of a cube (default at N=1: 512 = BASELINE configs[2]; N>1 without --size/--grid: "
                         "1024x1024x512 = configs[3])")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--cpu-size", type=int, default=256, help="C2 of BASELINE.md section 3")
    ap.add_argument("--cpu-steps", type=int, default=8, help="timed steps of the C2 CPU baseline (256^3: about 0.5 s per step on 16 cores)")
    ap.add_argument("--cpu-c1-steps", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--body", default="sphere", choices=["sphere", "donut", "cylinder", "tgv"],
                    help="donut: BASELINE configs[4] (use with --dtype f64); cylinder: moving body, measure! every step (use with "
                         "--dtype f64: in Float32 the C restatement of the reference's solver stalls on it from 256^3 -- not "
                         "verified on WaterLily.jl, DESIGN.md section 5); tgv: no body")
    ap.add_argument("--layout", default="padded", choices=["padded", "dense"],
                    help="padded: rows 128-B aligned (default); dense: the reference's column-major layout (pitch N+2)")
    ap.add_argument("--no-dense-leg", action="store_true", help="N=1: skip the extra 5-step run in the dense (drop-in) layout")
    ap.add_argument("--kernel", default=None, help="force the kernel class reported in `roofline`")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "host", "loopback"],
                    help="multi-rank transport: rccl (one GPU per rank; a failing communicator is a non-zero exit), host "
                         "(explicit: gloo staging; lets ranks share a GPU, tests -- never an xGMI number) or loopback (ONE process "
                         "plays rank --rank of --gpus: every exchange is a device copy of its own planes, the per-rank compute "
                         "time of an N-GPU run measured on one GPU)")
    ap.add_argument("--rank", type=int, default=None, help="--comm loopback: which rank's slab to run (default: the middle one)")
    ap.add_argument("--grid", type=int, nargs=3, default=None,
                    help="explicit GLOBAL grid nx ny nz (strong scaling, e.g. 1024 1024 512 = BASELINE configs[3])")
    ap.add_argument("--weak", action="store_true",
                    help="N>1: weak scaling, every GPU keeps a size^3 slab (global size x size x size*N) instead of the fixed global grid")
    ap.add_argument("--no-ref1", action="store_true",
                    help="N>1 strong scaling: do not time the same global grid on ONE GPU (rank 0, after the N-GPU run) for `speedup_vs_1gpu`")
    ap.add_argument("--ref1-steps", type=int, default=5)
    return ap.parse_args(argv)


def self_launch_cmd(ngpus, port, argv):
    """the launcher line of the bench contract: one rank per GPU under torch.distributed.run"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as CHILD processes, relay what they
    print (rank 0's JSON line) and leave with their status.  This parent never imports torch and never touches a GPU (a
    process that has initialised the GPU must not exec / fork GPU work on this pool), and it does not exec."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(16, (os.cpu_count() or 1)) // max(1, args.gpus))))
    proc = subprocess.Popen(self_launch_cmd(args.gpus, port, argv), env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:
        if ln.startswith('{"metric"'):
            line = ln
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        sys.stdout.write(line)
        sys.stdout.flush()
    if rc == 0 and line is None:
        print("[bench] the ranks finished without printing a result line", file=sys.stderr)
        rc = 4
    return rc


def make_sim(args, dims, T, dev, padded, fit=None):
    make = {"sphere": sphere, "donut": donut, "cylinder": moving_cylinder, "tgv": tgv}[args.body]
    if fit is not None and args.body in ("sphere", "donut"):
        return make(dims, T, device=dev, padded=padded, fit=fit)
    return make(dims, T, device=dev, padded=padded)


def timed_steps(sim, steps, warmup, remeasure, sync):
    """`warmup` untimed steps, then `steps` steps between two synchronisations: seconds"""
    from waterlily_amd import sim as S
    for _ in range(warmup):
        S.sim_step(sim, remeasure=remeasure)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        S.sim_step(sim, remeasure=remeasure)
    sync()
    return time.perf_counter() - t0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and args.comm != "loopback":
        raise SystemExit(self_launch(args, argv))          # (before torch is imported: the parent stays off the GPU)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    loopback = args.comm == "loopback"
    if loopback:
        if world != 1:
            raise SystemExit("--comm loopback is a one-process run")
        world = max(1, args.gpus)
        rank = (min(1, world - 1) if args.rank is None else args.rank)      # default: the second rank (two neighbours, no z boundary)
    elif world != max(1, args.gpus):
        raise SystemExit(f"bench.py --gpus {args.gpus} was started by a launcher with WORLD_SIZE={world}")

    from waterlily_amd import _lib
    from waterlily_amd import dist as wd
    from waterlily_amd import sim as S
    L = _lib.lib()
    T = np.float32 if args.dtype == "f32" else np.float64
    tsz = np.dtype(T).itemsize
    transport = "none" if world == 1 else {"rccl": "rccl", "host": "host-staging(gloo)", "loopback": "loopback(device copies, one process)"}[args.comm]
    if args.comm == "rccl" and os.environ.get("WL_RCCL_OVER_SOCKETS") == "1" and world > 1:
        transport = "rccl(socket transport between ranks sharing one GPU: a test, not xGMI)"
    # WL_RCCL_OVER_SOCKETS=1 (tests on a ONE-GPU box): the ranks share the device and present themselves to RCCL as different hosts,
    # so the library's RCCL communicator pairs them over its socket transport -- every RCCL call of the run really executes between
    # processes; torch's own process group (barriers, the max over the ranks) then runs on gloo.  Never an xGMI number.
    sockets = real_rccl_sockets = (args.comm == "rccl" and os.environ.get("WL_RCCL_OVER_SOCKETS") == "1" and world > 1)
    if sockets:
        os.environ.update(NCCL_HOSTID=f"wl-bench-host-{rank}", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_P2P_DISABLE="1",
                          NCCL_SHM_DISABLE="1", NCCL_NET_GDR_LEVEL="0")
    if args.comm in ("host", "loopback") or sockets:
        local = local % max(1, torch.cuda.device_count())
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)
    real = world > 1 and not loopback                 # several processes
    if real and args.comm == "rccl":
        # one process per GPU; the z axis is cut into `world` slabs, halos + scalar all-reduces run over RCCL (xGMI).
        # A communicator that cannot be created is a failed run (non-zero exit with the library's error text): a number
        # produced over any other transport would not be an xGMI measurement.
        if real_rccl_sockets:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev))
        try:
            wd.init_rccl()
        except Exception as e:
            print(f"[bench] rank {rank}: RCCL communicator failed: {e} | {L.wl_last_error().decode()}", file=sys.stderr, flush=True)
            os._exit(3)   # (peers still inside ncclCommInitRank leave through dist.init_rccl's watchdog, WL_COMM_TIMEOUT, or the launcher)
    elif real:
        dist.init_process_group("gloo")
        wd.init_host()
    elif loopback and world > 1:
        wd.init_loopback(rank, world)
    cr, cn = C.c_int(), C.c_int()
    _lib.check(L.wl_comm_rank(C.byref(cr), C.byref(cn)))
    if cn.value != world:
        raise SystemExit(f"libwlhip communicator has {cn.value} ranks, launcher has {world}")
    if real and args.comm == "rccl" and wd.kind() != "rccl":
        raise SystemExit("bench.py --comm rccl: the library's communicator is not RCCL")
    # Workloads.  N=1: the BASELINE 512^3 cube (configs[2]).  N>1: STRONG scaling on BASELINE configs[3], 1024 x 1024 x 512
    # (the north star's ">= 6x at 8 GPUs vs 1 GPU" case), unless --size (a cube, strong) / --grid (explicit) / --weak is given.
    if args.grid:
        dims, scaling = tuple(args.grid), "strong"
    elif args.weak:
        m = args.size or 512
        dims, scaling = (m, m, m * world), "weak"
    elif args.size or world == 1:
        m = args.size or 512
        dims, scaling = (m, m, m), ("strong" if world > 1 else "weak")
    else:
        dims, scaling = C4_GRID, "strong"
    fit = None
    if loopback and world > 1:
        # A rank in loopback takes its neighbours to be copies of itself: the run is a periodic stack of THIS slab.  A body that
        # crossed the seam would end there (the solver stalls on it, tools/loopback_ranks.py) and a slab without any body is the
        # trivial uniform stream (pcg! leaves at once): the body is shrunk to fit the slab and centred in it -- the same kernels
        # over the same cells, a full solve, a busy-row share of the same order.
        nzl = dims[2] // world
        fit = (rank * nzl, (rank + 1) * nzl)
    sim = make_sim(args, dims, T, dev, args.layout == "padded", fit)
    remeasure = args.body == "cylinder"        # the moving body is re-measured every step (sim_step!'s default)
    ncell_global = int(np.prod(dims))
    ncell = ncell_global // world            # cells per rank: threshold for "finest level" launches
    names = class_table(L)
    if args.kernel is not None and (args.kernel not in names or args.kernel not in ALG_T):
        raise SystemExit(f"--kernel {args.kernel}: not a priced kernel class (choose from {sorted(k for k in ALG_T if k in names)})")

    def sync():
        torch.cuda.synchronize()
        if real:
            dist.barrier()

    def timed_class(nm):
        """one extra step with every finest-level launch of class nm bracketed by hipEvents"""
        _lib.check(L.wl_prof_reset())
        _lib.check(L.wl_prof_select(names[nm], int(0.9 * ncell)))
        S.sim_step(sim, remeasure=remeasure)
        nl, nc, ms = C.c_int64(), C.c_int64(), C.c_double()
        _lib.check(L.wl_prof_timed(C.byref(nl), C.byref(nc), C.byref(ms)))
        _lib.check(L.wl_prof_select(-1, 0))
        return {"launches": nl.value, "ms": ms.value, "cells": nc.value}

    for w in range(args.warmup):
        S.sim_step(sim, remeasure=remeasure)
    sync()
    # each finest-level class is timed in one extra warm-up step: finds the dominant kernel, feeds the smoother objects
    classes = ["pcg_mult_dot", "pcg_update", "pcg_direction", "smooth", "prolongate", "conv_diff", "bdim", "residual", "correct",
               "div", "cfl", "scale"]
    per_class = {nm: timed_class(nm) for nm in classes if nm in names}
    dominant = args.kernel or max(per_class, key=lambda k: per_class[k]["ms"])

    # timed region: only the dominant class is bracketed by hipEvents (on the library's stream)
    _lib.check(L.wl_prof_reset())
    _lib.check(L.wl_prof_select(names[dominant], int(0.9 * ncell)))
    n0 = len(sim.pois.n)
    sync()
    t0 = time.perf_counter()
    for it in range(args.steps):
        if it == args.steps - 1 and world > 1:
            _lib.check(L.wl_prof_reset_comm())   # count the collectives of the last timed step only (host counters: no GPU work)
        S.sim_step(sim, remeasure=remeasure)
    sync()
    elapsed = time.perf_counter() - t0
    if real:  # MAX over ranks
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if (args.comm == "rccl" and not real_rccl_sockets) else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    nl, nc, ms = C.c_int64(), C.c_int64(), C.c_double()
    _lib.check(L.wl_prof_timed(C.byref(nl), C.byref(nc), C.byref(ms)))
    _lib.check(L.wl_prof_select(-1, 0))
    vcycles = sim.pois.n[n0:]

    mlups = ncell_global * args.steps / elapsed / 1e6
    n_uni, n_rows = S.uniform_rows(sim.pois, 0)
    phi = n_uni / max(1, n_rows)                       # share of x-rows whose L / iD loads are skipped
    cube = dims[0] == dims[1] == dims[2]
    traffic_db = {}
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile) and cube and world == 1 and args.layout == "padded":
        try:
            traffic_db = json.load(open(tfile))
        except Exception:
            traffic_db = {}

    tkey = f"{dims[0]}^3/{args.dtype}" + ("" if args.body == "sphere" else "/" + args.body)   # case key of profiles/traffic.json

    def kernel_record(nm, launches, cells, total_ms):
        """bytes / rates of one kernel class from its launch count, summed cells and summed duration"""
        if not launches or total_ms <= 0:
            return None
        dense, need, extra = ALG_T[nm]
        avg_ms = total_ms / launches
        cpl = cells / launches
        need_b = (need + (1.0 - phi) * extra) * tsz * cpl
        dense_b = dense * tsz * cpl
        rate = need_b / (avg_ms * 1e-3) / 1e9
        drate = dense_b / (avg_ms * 1e-3) / 1e9
        rec = {"bound": "hbm", "kernel": nm, "achieved": rate, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rate / HBM_PEAK_GBS,
               "traffic": traffic_db.get(f"{nm}@{tkey}"), "launches": launches, "avg_launch_ms": avg_ms,
               "algorithmic_bytes_per_launch": need_b,
               "dense": {"bytes_per_launch": dense_b, "GB/s": drate, "frac_of_peak": drate / HBM_PEAK_GBS,
                         "note": "SURVEY 8(d) per-cell figure of the reference operator; the fused kernel does not move these bytes"}}
        if rec["traffic"]:
            rec["traffic_GB/s"] = rec["traffic"] / (avg_ms * 1e-3) / 1e9
            # a committed constant of an earlier rocprofv3 --pmc run of this command (tools/profile.sh), not measured by THIS run
            rec["traffic_source"] = "profiles/" + str(traffic_db.get("_source", {}).get(tkey, "?")) + "_pmc_traffic_*"
            rec["traffic_build_matches"] = traffic_db.get("_csrc_sha1", {}).get(tkey) == _csrc_digest()   # same kernel sources as profiled?
        return rec

    roof = kernel_record(dominant, nl.value, nc.value, ms.value)
    if roof is None:   # the class had no finest-level launch in the timed region: say so instead of failing after the run
        roof = {"bound": "hbm", "kernel": dominant, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                "traffic": None, "launches": int(nl.value), "note": "no finest-level launch of this class was timed"}
    roof["uniform_row_fraction"] = phi
    # (one extra step per class right after warm-up: a pcg! call that leaves early still enqueues its remaining kernels as
    #  no-ops, so ms / launches here is NOT a per-kernel duration -- `avg_launch_ms` above and profiles/ are)
    roof["per_class_ms_one_step"] = {k: {"launches": v["launches"], "ms": v["ms"]} for k, v in per_class.items()}
    sm = per_class.get("smooth")
    pr = per_class.get("prolongate")
    Re = {"sphere": 3700, "tgv": 1600}.get(args.body, 1000)
    what = ("uniform inflow, remeasure=false" if args.body in ("sphere", "donut") else
            ("body moving through fluid at rest, remeasure=true (native measure! + changed-rows update! every step)" if remeasure else "no body"))
    tag = ""
    if args.dtype == "f32" and args.body == "sphere":
        if dims == (512, 512, 512):
            tag = " (BASELINE configs[2])"
        elif dims == C4_GRID:
            tag = " (BASELINE configs[3])"
        elif dims == (256, 256, 256):
            tag = " (BASELINE configs[1])"
    if args.dtype == "f64" and args.body == "donut" and dims == (512, 512, 512):
        tag = " (BASELINE configs[4])"
    workload = (f"3D {args.body} {dims[0]}x{dims[1]}x{dims[2]}, Re={Re}, {args.dtype}, {what}" + (", dense layout" if args.layout == "dense" else "") + tag
                + ("" if world == 1 else (f", rank {rank} of {world} z-slabs alone on one GPU (loopback exchanges)" if loopback else f", z-slabs over {world} GPUs")))
    out = {
        "metric": "MLUPS (cell-updates/s) per sim_step!, 3D sphere", "value": mlups, "unit": "MLUPS",   # (loopback: N x this rank's rate)
        "n_gpus": 1 if loopback else world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": workload, "layout": args.layout,
                   "transport": transport, "comm_ranks": cn.value,
                   "scalar_allreduce": ("none" if world == 1 else ("mailbox(pinned host memory)" if wd.mailbox_active() else
                                                                   ("ncclAllReduce" if args.comm == "rccl" else ("device scaling" if loopback else "host callbacks")))),
                   "collectives_last_step": S.comm_counts() if world > 1 else None,
                   "vcycles_per_solve": vcycles[:6], "mean_vcycles_per_step": float(np.sum(vcycles)) / args.steps},
        "roofline": roof,
        "smoother": kernel_record("smooth", sm["launches"], sm["cells"], sm["ms"]) if sm else None,
        "prolong_increment": kernel_record("prolongate", pr["launches"], pr["cells"], pr["ms"]) if pr else None,
    }
    if loopback and world > 1:
        # value = what `world` such ranks would deliver if nothing but this rank's own work limited them (no wire time)
        stalled = max(vcycles) >= 32 if vcycles else False
        out["loopback"] = {"rank": rank, "of": world, "per_rank_ms_per_step": None if stalled else out["ms_per_step"],
                           "solver_converged": not stalled,
                           "caveat": ("the solver STALLED (32 V-cycles per solve): with more than two ranks the replicated coarse levels -- the "
                                      "global problem with its real walls -- do not fit the slab's copy-of-itself neighbours; only the per-launch "
                                      "times of the kernel classes are meaningful in this line, not ms_per_step") if stalled else None,
                           "body_fit_to_slab_planes": list(fit) if fit else None,
                           "note": "one process plays one rank of the N-GPU run: its slab, its split launches, its reductions; every "
                                   "exchange is a device copy, every all-reduce a scaling kernel -- compute + launch time of a rank, no wire; "
                                   "the body is shrunk to fit the slab (a periodic stack of this slab is what the run solves)"}
    del sim
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    if world == 1 and not args.no_dense_leg and args.layout == "padded":
        # the layout the reference-side binding hands over (julia/WaterLilyHIPNativeExt.jl: dense column-major, pitch N+2)
        try:
            sd = make_sim(args, dims, T, dev, False)
            k, wd_ = min(5, args.steps), 10        # (10 untimed steps first: the start-up steps take 2-3 V-cycles per solve)
            td = timed_steps(sd, k, wd_, remeasure, sync)
            out["layout_dense"] = {"ms_per_step": td / k * 1e3, "value": ncell_global * k / td / 1e6, "unit": "MLUPS", "steps": k, "warmup": wd_,
                                   "vcycles_per_solve": sd.pois.n[-6:], "vs_padded": (td / k) / (elapsed / args.steps),
                                   "note": "the reference's dense strides (pitch N+2); the shim's HIPArray allocates pitched rows instead"}
            del sd
            gc.collect()
            torch.cuda.empty_cache()
        except Exception as e:                                    # the headline number stands on its own
            out["layout_dense"] = {"error": str(e)[:200]}
    if real:
        wd.finalize()
        dist.destroy_process_group()          # (before rank 0's one-GPU leg: nobody waits in a collective for it)
    if real and scaling == "strong" and not args.no_ref1 and rank == 0:
        # the same global grid on ONE GPU (rank 0's, the communicator is gone): the denominator of the strong-scaling claim
        try:
            s1 = make_sim(args, dims, T, dev, args.layout == "padded")
            k = max(1, args.ref1_steps)
            t1 = timed_steps(s1, k, 8, remeasure, torch.cuda.synchronize)     # (8 untimed steps: past the start-up V-cycles)
            out["one_gpu"] = {"ms_per_step": t1 / k * 1e3, "value": ncell_global * k / t1 / 1e6, "unit": "MLUPS", "steps": k, "warmup": 8,
                              "vcycles_per_solve": s1.pois.n[-6:],
                              "note": "same grid, same build, one GPU (rank 0 after the N-GPU run)"}
            out["speedup_vs_1gpu"] = (t1 / k) / (elapsed / args.steps)
            del s1
        except Exception as e:
            out["one_gpu"] = {"error": str(e)[:200]}
            out["speedup_vs_1gpu"] = None
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args.cpu_size, args.cpu_steps, args.cpu_c1_steps)
    if rank == 0 or loopback:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
