"""Worker for tests/test_multi_gpu.py: launched with torch.distributed.run, N ranks sharing ONE GPU.

Every rank builds (a) the undecomposed simulation and (b) its z-slab of the decomposed one, steps both and
compares the gathered slab fields with the undecomposed ones.  Transport: torch.distributed gloo through the
host-callback communicator (the RCCL communicator needs one GPU per rank; the C++ code above the transport
-- slab kernels, halo exchange placement, all-reduces, replicated coarse levels -- is identical)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C  # noqa: E402

from waterlily_amd import _lib  # noqa: E402
from waterlily_amd import dist as wd  # noqa: E402
from waterlily_amd import sim as S  # noqa: E402
from waterlily_amd.body import AutoBody, norm2  # noqa: E402


def main():
    """argv[1]: one case, or several joined by '+' (same transport: they share the process group and the communicator, so the
    start-up of the ranks -- most of a small case's wall time -- is paid once); rank 0 prints one `RESULT <json>` line per case"""
    cases = (sys.argv[1] if len(sys.argv) > 1 else "sphere_f32").split("+")
    case = cases[0]
    if "rcclnet" in case:
        # RCCL itself at N > 1 on a ONE-GPU box: every rank claims to sit on a different host (NCCL_HOSTID), so RCCL pairs the
        # ranks over its socket transport (loopback) instead of refusing two ranks on one device.  Not an xGMI path, but the
        # library's whole RCCL call pattern -- ncclCommInitRank, ncclCommSplit, grouped ncclSend/ncclRecv on the comm stream,
        # ncclAllReduce / ncclAllGather on the compute stream -- really executes between two processes.
        rk = os.environ.get("RANK", "0")
        os.environ.update(NCCL_HOSTID=f"wl-test-host-{rk}", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_P2P_DISABLE="1",
                          NCCL_SHM_DISABLE="1", NCCL_NET_GDR_LEVEL="0")
        torch.cuda.set_device(0)
        dist.init_process_group("gloo")
        wd.init_rccl()
    elif "rccl" in case:   # the production transport: one GPU per rank (world size 1 on a 1-GPU box)
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        wd.init_rccl()
    else:
        dist.init_process_group("gloo")
        wd.init_host()
    rank, size = dist.get_rank(), dist.get_world_size()
    for case in cases:
        out = run_case(case, rank, size)
        if out is not None and rank == 0:
            print("RESULT " + json.dumps(dict(out, case=case)), flush=True)
        dist.barrier()
    if any(c.startswith("mboxtimeout") for c in cases):
        _lib.lib().wl_comm_finalize()       # (the mailbox's error flag is sticky: the return code says so again)
    else:
        wd.finalize()
    dist.destroy_process_group()


def run_case(case, rank, size):
    T = np.float64 if case.endswith("f64") else np.float32
    m = 512 if "big" in case else (40 if "five" in case else 32)     # (40 = 5 * 2^3: five slabs of 8, 4, 2 planes)
    dims = (m, m, m) if "long" not in case else ((2 * m, 2 * m, 4 * m) if "vlong" in case else (m, m, 2 * m))
    R, c = m / 8, m / 2 - 1
    if case.startswith("donut"):
        Rm, rm, cc = m / 4, m / 16, m / 2

        def sdf(x, t):
            q = torch.sqrt((x[1] - cc) ** 2 + (x[2] - cc) ** 2) - Rm
            return torch.sqrt((x[0] - cc) ** 2 + q ** 2) - rm
        body, L, nu = AutoBody(sdf), Rm, Rm / 1000
    elif "move" in case:   # parametric sphere drifting along z across the slab boundaries: native measure! kernels + the
        from waterlily_amd import body as B                      # changed-rows update!(pois) every step
        body, L, nu = B.Sphere(c, R, 3, map=B.translation(3, v=(0.2, 0.0, 0.9))), 2 * R, 2 * R / 500
    elif "big" in case:    # (parametric sphere: the closure + autograd measure of 134 M cells on the host takes a minute per rank)
        from waterlily_amd import body as B
        body, L, nu = B.Sphere(c, R, 3), 2 * R, 2 * R / 3700
    else:
        body, L, nu = AutoBody(lambda x, t: norm2(x - c) - R), 2 * R, 2 * R / 3700
    perdir = ()
    if "zper" in case:
        perdir = (2,)
    if "yzper" in case:
        perdir = (1, 2)
    if "yper" in case:
        perdir = (1,)
    g = (lambda i, t: 0.05 * t if i == 0 else 0.0) if "accel" in case else None
    kw = dict(nu=nu, body=body, T=T, exitBC=("exit" in case), perdir=perdir, g=g)
    if "rccl" in case and "rcclnet" not in case:
        kw["device"] = f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"
    if case.startswith("vtk"):
        return vtk_roundtrip(rank, size, dims, L, kw)
    if case.startswith("mboxtimeout"):
        return mailbox_timeout(rank, size)
    if case.startswith("arlat"):
        us = C.c_double()
        _lib.check(_lib.lib().wl_prof_allreduce_us(500, C.byref(us)))
        return {"us_per_allreduce": us.value, "mailbox": wd.mailbox_active(), "ranks": size}
    # "oblique": a stream with a large component along z leaves conv_diff!'s flux scratch ~ w^2 in sigma's ghost cells of the exit
    # plane, above every interior flux_out: the whole-array maximum(a.sigma) of CFL (Flow.jl:174) is a GHOST cell, found by the
    # shell reduction of whichever rank owns it
    ubc = (0.5, 1.0, 4.0) if "oblique" in case else (1.0, 0.0, 0.0)
    nov_c = C.c_int64()
    _lib.check(_lib.lib().wl_prof_overlapped(C.byref(nov_c)))
    nov0 = int(nov_c.value)                     # (the counter runs over the whole process: several cases may share it)
    ref = S.Simulation(dims, ubc, L, slab=None, **kw)
    slab = wd.Slab(rank, size, dims[2], ring=(2 in perdir))
    # "deep": keep every level a slab as long as the partition allows; default: replicate levels <= 2^21 cells
    sim = S.Simulation(dims, ubc, L, slab=slab, replicate_cells=0 if "deep" in case else 1 << 21, **kw)
    out = {"rank": rank, "levels": [(tuple(l.layout.Ng), l.layout.slab is not None) for l in sim.pois.levels]}
    big = "big" in case

    def on_device(a, b, rel):
        """max |slab - undecomposed| over the planes this rank owns, on the GPU, then the max over the ranks (the big case:
        gathering 512^3 fields through pickled host arrays takes a minute)"""
        sl = a._wl_slab
        own = a[:, :, sl.own_lo:sl.own_hi + 1]
        d = (own - b[:, :, sl.kz0 + sl.own_lo:sl.kz0 + sl.own_hi + 1]).abs().max().double().cpu()
        mx = b.abs().max().double().cpu() if rel else torch.ones((), dtype=torch.float64)
        v = torch.stack([d, mx])
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        return float(v[0] / max(1e-30, float(v[1])))

    # static fields after construction
    for k in ("u", "mu0", "mu1", "V"):
        if big:
            out["init_" + k] = on_device(getattr(sim.flow, k), getattr(ref.flow, k), False)
            continue
        out["init_" + k] = float(np.max(np.abs(S.gather(getattr(sim.flow, k)) - S.to_host(getattr(ref.flow, k)))))
    nsteps = 1 if "big" in case else (2 if "vlong" in case else 3)
    for _ in range(nsteps):
        S.sim_step(ref, remeasure="move" in case)
        S.sim_step(sim, remeasure="move" in case)
    out["n_ref"], out["n_slab"] = list(ref.pois.n), list(sim.pois.n)
    out["dt_ref"], out["dt_slab"] = list(ref.flow.dt), list(sim.flow.dt)
    sg = S.to_host(ref.flow.sigma)
    out["sigma_max_whole_over_inside"] = float(sg.max() / sg[S.inside(sg)].max())
    for k in ("u", "p", "f"):
        if big:
            out["d_" + k] = on_device(getattr(sim.flow, k), getattr(ref.flow, k), True)
            continue
        a, b = S.gather(getattr(sim.flow, k)), S.to_host(getattr(ref.flow, k))
        if 2 in perdir and k == "f":
            # f on the two z GHOST planes: the reference leaves partial flux sums there (conv_diff! never periodic-copies
            # f), a ring of slabs has no such planes (gather() fills them by wrapping): compare the interior planes
            a, b = a[:, :, 1:-1], b[:, :, 1:-1]
        out["d_" + k] = float(np.max(np.abs(a - b)) / max(1e-30, np.max(np.abs(b))))
    nov = C.c_int64()
    _lib.check(_lib.lib().wl_prof_overlapped(C.byref(nov)))
    out["overlapped"] = int(nov.value) - nov0
    out["force_ref"] = S.pressure_force(ref).tolist()
    out["force_slab"] = S.pressure_force(sim).tolist()
    # collectives of ONE more step, counted by the library (wl_prof_comm), and the V-cycle counts of its two solves
    _lib.check(_lib.lib().wl_prof_reset())
    S.mom_step(sim.flow, sim.pois)
    out["comm"] = S.comm_counts()
    out["mailbox"] = wd.mailbox_active()
    out["n_counted"] = sim.pois.n[-2:]
    out["slab_nzl"] = [l.layout.slab.nzl if l.layout.slab is not None else None for l in sim.pois.levels]
    return out


def mailbox_timeout(rank, size):
    """the mailbox all-reduce's waits are bounded: a rank whose peer never posts gives up, and the library reports it"""
    import time
    Lb = _lib.lib()
    assert wd.mailbox_active()
    v = (C.c_double * 1)(float(rank + 1))
    _lib.check(Lb.wl_allreduce(v, 1, 0))                       # a healthy round first
    ok_sum = v[0] == sum(range(1, size + 1))
    S.set_option(26, 1)                                        # one second of the device's wall clock
    raised, msg = False, ""
    if rank == 0:
        try:
            _lib.check(Lb.wl_allreduce(v, 1, 0))               # rank 1 never joins this one
        except _lib.WlError as e:
            raised, msg = True, str(e)
    else:
        time.sleep(4.0)
    return {"ok_sum": bool(ok_sum), "raised": raised, "msg": msg}


def vtk_roundtrip(rank, size, dims, L, kw):
    """VTK write -> restart on z-slabs (ext/WaterLilyWriteVTKExt.jl:57-66, ext/WaterLilyReadVTKExt.jl:28-45): the slabs
    are gathered and rank 0 writes; on restart every rank loads the file and keeps its slab.  The restarted decomposed
    simulation, and an undecomposed one restarted from the same file, must hold bitwise the fields that were written."""
    from waterlily_amd import vtk
    tmp = os.environ["WL_TMP"]
    mk = lambda sl: S.Simulation(dims, (1.0, 0.0, 0.0), L, slab=sl, **kw)
    sim = mk(wd.Slab(rank, size, dims[2]))
    for _ in range(2):
        S.sim_step(sim, remeasure=False)
    wr = vtk.vtkWriter(os.path.join(tmp, "slab_vtk"), dir=os.path.join(tmp, "SLAB_DIR"))
    vtk.write(wr, sim)
    vtk.close(wr)
    dist.barrier()
    pvd = os.path.join(tmp, "slab_vtk.pvd")
    again, whole = mk(wd.Slab(rank, size, dims[2])), mk(None)
    vtk.restart_sim(again, fname=pvd)
    vtk.restart_sim(whole, fname=pvd)
    out = {"rank": rank}
    for k in ("u", "p"):
        a, b, c = S.gather(getattr(sim.flow, k)), S.gather(getattr(again.flow, k)), S.to_host(getattr(whole.flow, k))
        out["same_" + k] = bool(np.array_equal(a, b)) and bool(np.array_equal(a, c))
    # the halo planes of the restarted slab are current too (the next conv_diff! reads two of them)
    out["same_local_u"] = bool(torch.equal(sim.flow.u, again.flow.u))
    out["dt"] = [sim.flow.dt[-1], again.flow.dt[-2], whole.flow.dt[-2]]
    out["cfl"] = [again.flow.dt[-1], whole.flow.dt[-1]]
    S.sim_step(again, remeasure=False)
    S.sim_step(whole, remeasure=False)
    a, b = S.gather(again.flow.u), S.to_host(whole.flow.u)
    out["d_next"] = float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
    out["n_next"] = [again.pois.n[-2:], whole.pois.n[-2:]]
    res = [None] * size
    dist.all_gather_object(res, out["same_local_u"])
    out["same_local_u"] = all(res)
    return out


if __name__ == "__main__":
    main()
