import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from waterlily_amd import sim as S, body as B
sim = bench.sphere((512,)*3, np.float32)
dims = tuple(n-2 for n in sim.flow.N)
cand = sim.flow._band_cells[1]
print("candidates", cand.numel())
def tm(name, fn, n=3):
    for i in range(n):
        torch.cuda.synchronize(); t0=time.perf_counter(); r=fn(); torch.cuda.synchronize()
        print(f"{name} {i}: {(time.perf_counter()-t0)*1e3:.2f} ms")
    return r
idx,nds = tm("nds_band_from_candidates", lambda: B.nds_band_from_candidates(sim.body, dims, cand, t=0.0))
print(idx.numel())
tm("band_to_device_t", lambda: S.band_to_device_t(sim.flow.p, idx, nds))
# pieces
D=3; Ng=tuple(n+2 for n in dims); strides=np.cumprod((1,)+Ng[:-1])
def pts_():
    rem, coords = cand.clone(), []
    for ddim in range(D-1,-1,-1):
        coords.insert(0, rem // int(strides[ddim])); rem = rem % int(strides[ddim])
    return torch.stack([c.to(torch.float64)-0.5 for c in coords])
pts = tm("points", pts_)
tm("measure", lambda: B.measure(sim.body, pts, 0.0, fastd2=1.0))
