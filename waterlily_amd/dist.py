"""Multi-GPU bootstrap: z-slab decomposition, one process per GPU.

The data path (halo planes, scalar all-reduces, coarse-level all-gather) lives in libwlhip.so and runs over RCCL
(xGMI) on the compute stream.  This module only (a) computes the slab partition, (b) bootstraps the RCCL
communicator by broadcasting the 128-byte unique id through torch.distributed, and (c) provides the *host-callback*
communicator used by tests, where the same C++ code paths are driven over torch.distributed `gloo` (several ranks
may then share one GPU, or -- for the pure host-logic tests -- no GPU at all).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib

HZ = 2  # halo depth of a decomposed array: QUICK reads I-2..I+1 across a face (src/Flow.jl:6)


@dataclass
class Slab:
    """Partition of the z axis (interior extent nz) over `size` ranks for ONE grid level."""
    rank: int
    size: int
    nz: int                      # global interior planes of this level
    ring: bool = False           # z periodic across the slabs: ranks form a ring, nobody owns the z ghost planes

    def __post_init__(self):
        if self.nz % self.size:
            raise ValueError(f"z extent {self.nz} is not divisible by {self.size} ranks")
        self.nzl = self.nz // self.size           # interior planes per rank
        self.n2l = self.nzl + 2 * HZ              # local planes incl. halo
        self.nzg = self.nz + 2                    # global planes incl. the ghost layer
        self.kz0 = self.rank * self.nzl + 1 - HZ  # global index of local plane 0
        self.own_lo = HZ - (1 if (self.rank == 0 and not self.ring) else 0)     # rank 0 owns the lower ghost plane
        self.own_hi = HZ + self.nzl - 1 + (1 if (self.rank == self.size - 1 and not self.ring) else 0)

    def coarser(self) -> Optional["Slab"]:
        """The slab of the next multigrid level, or None when that level must be replicated
        (children of a coarse cell must live on one rank, and a slab needs >= 2 planes)."""
        if self.nzl % 2 or self.nzl // 2 < 2:
            return None
        return Slab(self.rank, self.size, self.nz // 2, self.ring)


def divisible(shape) -> bool:
    """MultiLevelPoisson.jl:36-37: the reference tests size(l.x), i.e. the extents INCLUDING the ghost layer"""
    return all(n % 2 == 0 and n > 4 for n in shape)


def plan_levels(Ng, slab: Optional[Slab], maxlevels: int = 10, replicate_cells: int = 1 << 21):
    """The multigrid hierarchy of `MultiLevelPoisson` (restrictML, MultiLevelPoisson.jl:18-25,53-55) for arrays of
    UNDECOMPOSED extents `Ng` (ghosts included): a list of (Ng_level, slab_or_None).  Multi-GPU: a level stays a z-slab
    while every rank keeps >= 2 (even) planes AND the level has more than `replicate_cells` interior cells; from the
    first level that fails either test on, every level is REPLICATED on all ranks (all-gather at the hand-over): no halo
    exchange and no all-reduce per dot product on the latency-bound levels, and the hierarchy -- hence pois.n -- is
    exactly the single-device one.  (Pure host arithmetic: tests/test_multi_gpu.py checks the 8-rank tables of C4 / C5.)"""
    Ng = tuple(int(n) for n in Ng)
    out = [(Ng, slab)]
    while divisible(out[-1][0]) and len(out) <= maxlevels:
        Na = tuple(1 + n // 2 for n in out[-1][0])
        slab = slab.coarser() if slab is not None else None
        if slab is not None and int(np.prod([n - 2 for n in Na])) <= replicate_cells:
            slab = None
        out.append((Na, slab))
    return out


def collectives_per_step(levels, vcycles, exitBC: bool = False, pcg_it: int = 6):
    """How many collectives one `mom_step!` issues on a rank of a z-slab run, from the level plan and the V-cycle counts of
    its two solves (`pois.n[-2:]`) -- the model DESIGN.md section 6 prices an 8-GPU step with, asserted against the
    library's own counters (wl_prof_comm) in tests/test_multi_gpu.py.
      all-reduces: CFL 1; exitBC! 1 (Flow.jl:160, once per step); per solve: residual! 1 + per V-cycle
                   [(1 + 2*it) per slab level (pcg!: rho, then z.eps and r.z' / r.r per iteration) + 1 (L2)];
      all-gathers: 1 per V-cycle (hand-over of the restricted residual to the replicated levels), if any level is replicated
                   while level 0 is a slab.
    Returns dict(allreduce=..., allgather=...)."""
    nslab = sum(1 for _, s in levels if s is not None)
    if nslab == 0:
        return {"allreduce": 0, "allgather": 0}
    handover = 1 if nslab < len(levels) else 0
    nv = int(sum(vcycles))
    ar = 1 + (1 if exitBC else 0) + len(vcycles) * 1 + nv * (nslab * (1 + 2 * pcg_it) + 1)
    return {"allreduce": ar, "allgather": nv * handover}


_state = {"kind": None, "rank": 0, "size": 1, "keep": None}


def rank_size():
    return _state["rank"], _state["size"]


def init_rccl() -> None:
    """Bootstrap the RCCL communicator inside libwlhip.so (torch.distributed must be initialised).

    ncclCommInitRank blocks until EVERY rank has joined: if one rank dies before it gets there the others would wait for
    ever.  A watchdog thread bounds that wait (WL_COMM_TIMEOUT seconds, default 300, 0 = none): the process then prints the
    reason and exits with code 5 instead of hanging -- under any launcher, not only one that reaps its ranks."""
    import os
    import sys
    import threading

    import torch.distributed as dist
    L = _lib.lib()
    rank, size = dist.get_rank(), dist.get_world_size()
    buf = (C.c_char * 128)()
    if rank == 0:
        _lib.check(L.wl_comm_unique_id(buf))
    obj = [bytes(buf)]
    dist.broadcast_object_list(obj, src=0)
    raw = (C.c_char * 128).from_buffer_copy(obj[0])
    limit = float(os.environ.get("WL_COMM_TIMEOUT", "300"))

    def give_up():
        print(f"[waterlily_amd] rank {rank}: RCCL communicator of {size} ranks not complete after {limit:.0f} s "
              "(a peer rank failed before joining?): exiting", file=sys.stderr, flush=True)
        os._exit(5)
    dog = threading.Timer(limit, give_up) if limit > 0 else None
    if dog:
        dog.daemon = True
        dog.start()
    try:
        _lib.check(L.wl_comm_init_rccl(raw, rank, size))       # (ctypes releases the GIL: the watchdog can fire meanwhile)
    finally:
        if dog:
            dog.cancel()
    _state.update(kind="rccl", rank=rank, size=size)
    init_mailbox()


def host_callbacks(group=None):
    """ctypes callbacks that carry libwlhip's collectives over torch.distributed on HOST buffers."""
    import torch
    import torch.distributed as dist
    rank, size = dist.get_rank(group), dist.get_world_size(group)

    def view(ptr, nbytes):
        return torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(nbytes,)))

    def sendrecv(user, slo, rlo, shi, rhi, nbytes, plo, phi):
        try:
            # order prescribed by include/wlhip.h (pairs correctly when plo == phi, a 2-rank periodic ring)
            ops = []
            if shi:
                ops.append(dist.P2POp(dist.isend, view(shi, nbytes), phi, group))
            if rlo:
                ops.append(dist.P2POp(dist.irecv, view(rlo, nbytes), plo, group))
            if slo:
                ops.append(dist.P2POp(dist.isend, view(slo, nbytes), plo, group))
            if rhi:
                ops.append(dist.P2POp(dist.irecv, view(rhi, nbytes), phi, group))
            for w in (dist.batch_isend_irecv(ops) if ops else []):
                w.wait()
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print("wl host sendrecv failed:", e, flush=True)
            return 1

    def allreduce(user, vals, n, op):
        try:
            t = torch.from_numpy(np.ctypeslib.as_array(vals, shape=(n,)))
            dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX, group=group)
            return 0
        except Exception as e:
            print("wl host allreduce failed:", e, flush=True)
            return 1

    def allgather(user, buf, nbytes):
        try:
            full = view(buf, nbytes * size)
            mine = full[rank * nbytes:(rank + 1) * nbytes].clone()
            dist.all_gather_into_tensor(full, mine, group=group)
            return 0
        except Exception as e:
            print("wl host allgather failed:", e, flush=True)
            return 1

    return _lib.SENDRECV_FN(sendrecv), _lib.ALLREDUCE_FN(allreduce), _lib.ALLGATHER_FN(allgather), rank, size


def init_host(group=None) -> None:
    """Install the host-callback communicator (tests; transport = whatever backend `group` uses, e.g. gloo)."""
    sr, ar, ag, rank, size = host_callbacks(group)
    _lib.check(_lib.lib().wl_comm_init_host(rank, size, sr, ar, ag, None))
    _state.update(kind="host", rank=rank, size=size, keep=(sr, ar, ag))
    init_mailbox(group)


def init_loopback(rank: int, size: int) -> None:
    """ONE process plays rank `rank` of a `size`-way z-slab run (include/wlhip.h: wl_comm_init_loopback): measurement of a rank's
    own compute and launch time on one GPU (bench.py --comm loopback).  No torch.distributed, no mailbox."""
    _lib.check(_lib.lib().wl_comm_init_loopback(int(rank), int(size)))
    _state.update(kind="loopback", rank=int(rank), size=int(size))


def kind():
    """"rccl", "host", "loopback" or None: the communicator this process installed"""
    return _state["kind"]


def init_mailbox(group=None) -> bool:
    """Switch the scalar all-reduces of the run to the library's mailbox (include/wlhip.h: wl_comm_mailbox): rank 0 creates a
    POSIX shared-memory object, every rank of the node maps it.  WL_MAILBOX=0 in the environment keeps ncclAllReduce / the
    host callbacks.  Returns whether the mailbox is active (a rank that cannot map it disables it everywhere)."""
    import os
    import uuid

    import torch.distributed as dist
    if os.environ.get("WL_MAILBOX", "1") == "0" or dist.get_world_size(group) < 2:
        return False
    L = _lib.lib()
    rank = dist.get_rank(group)
    name = [f"/wlhip-{os.getpid()}-{uuid.uuid4().hex[:12]}" if rank == 0 else None]
    dist.broadcast_object_list(name, src=0, group=group)
    ok = 1
    if rank == 0:
        ok = int(L.wl_comm_mailbox(name[0].encode(), 1) == 0)
    dist.barrier(group)
    if rank != 0:
        ok = int(L.wl_comm_mailbox(name[0].encode(), 0) == 0)
    oks = [None] * dist.get_world_size(group)
    dist.all_gather_object(oks, ok, group=group)
    if rank == 0:
        try:
            os.unlink("/dev/shm" + name[0])          # the mappings keep the memory alive; nothing is left behind in /dev/shm
        except OSError:
            pass
    if all(oks):
        # self-test before the run depends on it: one all-reduce of the rank numbers with a short bound on the waits; any
        # rank that sees a wrong sum or a time-out votes the mailbox off for everybody
        keep = C.c_int()
        _lib.check(L.wl_get_option(26, C.byref(keep)))       # (the caller's bound comes back after the test)
        L.wl_set_option(26, 10)
        v = (C.c_double * 1)(float(rank + 1))
        good = int(L.wl_allreduce(v, 1, 0) == 0 and v[0] == dist.get_world_size(group) * (dist.get_world_size(group) + 1) / 2)
        L.wl_set_option(26, keep.value)
        dist.all_gather_object(oks, good, group=group)
    if not all(oks):                                  # all or nothing: every rank must take the same path
        _lib.check(L.wl_comm_mailbox_off())
        if rank == 0:
            import sys
            print("[waterlily_amd] mailbox all-reduce unavailable on this node: scalars use the communicator's all-reduce",
                  file=sys.stderr, flush=True)
        return False
    return True


def mailbox_active() -> bool:
    on = C.c_int()
    _lib.check(_lib.lib().wl_comm_mailbox_active(C.byref(on)))
    return bool(on.value)


def finalize() -> None:
    _lib.check(_lib.lib().wl_comm_finalize())
    _state.update(kind=None, rank=0, size=1, keep=None)
