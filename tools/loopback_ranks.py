#!/usr/bin/env python3
"""Which ranks of a z-slab run can the loopback communicator (bench.py --comm loopback) stand in for?  Plays every rank of an
N-way split of a small sphere case in turn and prints the V-cycle counts next to those of the undecomposed run."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from waterlily_amd import _lib, dist as wd, sim as S  # noqa: E402

dims = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 64, 128)
P = int(sys.argv[4]) if len(sys.argv) > 4 else 8
ref = bench.sphere(dims, np.float32)
for _ in range(4):
    S.sim_step(ref, remeasure=False)
print("undecomposed", ref.pois.n, ref.flow.dt[-1])
del ref
for r in range(P):
    wd.init_loopback(r, P)
    s = bench.sphere(dims, np.float32)
    for _ in range(4):
        S.sim_step(s, remeasure=False)
    print("rank", r, s.pois.n, s.flow.dt[-1], flush=True)
    del s
    wd.finalize()
