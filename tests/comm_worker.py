"""N-process gloo check of the host-callback collectives (no GPU, no HIP call): the ctypes callbacks that carry
libwlhip's sendrecv / allreduce / allgather are invoked directly on host buffers."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waterlily_amd import dist as wd  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, size = dist.get_rank(), dist.get_world_size()
    sr, ar, ag, r, n = wd.host_callbacks()
    ok = (r, n) == (rank, size)
    # allreduce sum / max
    v = (C.c_double * 2)(rank + 1.0, 10.0 * (rank + 1))
    ok &= ar(None, v, 2, 0) == 0 and list(v) == [sum(range(1, size + 1)), 10.0 * sum(range(1, size + 1))]
    v = (C.c_double * 1)(float(rank))
    ok &= ar(None, v, 1, 1) == 0 and v[0] == size - 1
    # neighbour exchange: rank sends its id upward and downward
    nb = 64
    s_lo, s_hi = np.full(nb, rank, np.uint8), np.full(nb, 100 + rank, np.uint8)
    r_lo, r_hi = np.zeros(nb, np.uint8), np.zeros(nb, np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lo, hi = rank > 0, rank < size - 1
    ok &= sr(None, p(s_lo) if lo else None, p(r_lo) if lo else None, p(s_hi) if hi else None, p(r_hi) if hi else None, nb,
             rank - 1 if lo else -1, rank + 1 if hi else -1) == 0
    if lo:
        ok &= bool(np.all(r_lo == 100 + rank - 1))
    if hi:
        ok &= bool(np.all(r_hi == rank + 1))
    # periodic RING (z-periodic slabs): rank 0 and rank P-1 are neighbours.  With 2 ranks both peers are the same rank and
    # the prescribed op order must pair hi->lo and lo->hi
    plo, phi = (rank - 1) % size, (rank + 1) % size
    r_lo[:] = 0; r_hi[:] = 0
    ok &= sr(None, p(s_lo), p(r_lo), p(s_hi), p(r_hi), nb, plo, phi) == 0
    ok &= bool(np.all(r_lo == 100 + plo)) and bool(np.all(r_hi == phi))
    # in-place allgather
    buf = np.zeros(size * 8, np.uint8)
    buf[rank * 8:(rank + 1) * 8] = rank + 1
    ok &= ag(None, p(buf), 8) == 0 and bool(np.all(buf == np.repeat(np.arange(1, size + 1), 8)))
    res = [None] * size
    dist.all_gather_object(res, bool(ok))
    if rank == 0:
        print("RESULT " + json.dumps({"ok": all(res)}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
