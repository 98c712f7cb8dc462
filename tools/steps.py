#!/usr/bin/env python3
"""Run N steady-state steps of the 512^3 (or given) sphere case -- a clean target for rocprofv3 traces."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from waterlily_amd import sim as S  # noqa: E402
import torch  # noqa: E402
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
T = np.float64 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else np.float32
for kv in os.environ.get("WL_OPTS", "").split(","):     # WL_OPTS=27:0,30:0 -> wl_set_option before the run
    if ":" in kv:
        S.set_option(int(kv.split(":")[0]), int(kv.split(":")[1]))
sim = (bench.donut if os.environ.get('WL_BODY') == 'donut' else bench.sphere)((size,) * 3, T)
for _ in range(steps):
    S.sim_step(sim, remeasure=False)
torch.cuda.synchronize()
print("n:", sim.pois.n, "dt:", sim.flow.dt[-1])
