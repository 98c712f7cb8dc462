"""z-slab decomposition.

GPU test: 2 and 4 ranks share the one GPU of the box (transport: gloo through the host-callback communicator) and
must reproduce the undecomposed run: identical V-cycle counts, fields to a few ulp (only reduction grouping
differs: per-rank partial sums are combined by the all-reduce).

CPU tests (no GPU): the slab partition arithmetic and the host-callback collectives over a 2-process gloo group."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_workers(script, nproc, *args, timeout=600, **extra_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "tests", script), *args]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert lines, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(lines[-1][7:])


# ----------------------------------------------------------------------------- CPU: host logic

def test_slab_partition():
    from waterlily_amd.dist import HZ, Slab
    nz, P = 64, 4
    slabs = [Slab(r, P, nz) for r in range(P)]
    owned = []
    for s in slabs:
        assert s.n2l == nz // P + 2 * HZ
        owned += [s.kz0 + k for k in range(s.own_lo, s.own_hi + 1)]
    assert owned == list(range(nz + 2))          # every global plane (ghosts included) owned exactly once
    for a, b in zip(slabs[:-1], slabs[1:]):      # halo planes of a are b's first owned planes
        assert a.kz0 + a.own_hi + 1 == b.kz0 + b.own_lo
        assert a.own_hi + HZ <= a.n2l - 1 and b.own_lo - HZ >= 0
    # multigrid: levels stay slabs while >= 2 even planes per rank remain, then they are replicated
    s, chain = slabs[1], []
    while s is not None:
        chain.append(s.nzl)
        s = s.coarser()
    assert chain == [16, 8, 4, 2]
    with pytest.raises(ValueError):
        Slab(0, 3, 64)


def test_host_collectives_gloo_world2():
    out = run_workers("comm_worker.py", 2, timeout=120)
    assert out["ok"], out


# ----------------------------------------------------------------------------- GPU: decomposed == undecomposed

def check(out, T):
    tol = 2e-5 if T == "f32" else 1e-11
    assert out["n_ref"] == out["n_slab"], (out["n_ref"], out["n_slab"])
    assert np.allclose(out["dt_ref"], out["dt_slab"], rtol=tol)
    for k in ("init_u", "init_mu0", "init_mu1", "init_V"):
        assert out[k] == 0.0, (k, out[k])
    assert out["d_u"] < tol and out["d_f"] < tol and out["d_p"] < 20 * tol, out
    assert np.allclose(out["force_ref"], out["force_slab"], rtol=100 * tol, atol=100 * tol)


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,case", [(2, "sphere_deep_f32"), (4, "sphere_long_deep_f32"), (2, "donut_deep_f64"),
                                        (2, "sphere_exit_deep_f32"), (2, "sphere_f32"),
                                        (2, "sphere_zper_deep_f32"), (4, "sphere_long_zper_deep_f64"),
                                        (2, "sphere_yzper_accel_deep_f64"), (2, "sphere_yper_exit_accel_f32"),
                                        (2, "sphere_move_deep_f32"), (4, "sphere_long_move_f64")])
def test_slabs_match_undecomposed(nproc, case):
    out = run_workers("mg_worker.py", nproc, case)
    # "deep", 32^3 on 2 ranks: levels 32,16,8 (16,8,4 planes per rank) are slabs, 4^3 and 2^3 are replicated;
    # default: only the finest level is a slab (coarser ones hold <= 2^21 cells and are replicated)
    # z-periodic cases ("zper") run on a RING of slabs (rank 0 <-> rank P-1 exchange)
    nslab = sum(1 for _, d in out["levels"] if d)
    assert (nslab >= 3 if "deep" in case else nslab == 1) and not out["levels"][-1][1]
    assert out["overlapped"] > 0          # stencil launches were split around exchanges on the comm stream
    check(out, "f64" if case.endswith("f64") else "f32")


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,case", [(2, "sphere_deep_f32"), (4, "sphere_long_zper_deep_f64")])
def test_slabs_match_without_overlap(nproc, case):
    """Same, with the halo exchanges issued in-stream (WL_OVERLAP=0) instead of on the comm stream with the stencil
    launches split into inner planes (concurrent with the transfer) and the two boundary planes (after it)."""
    out = run_workers("mg_worker.py", nproc, case, WL_OVERLAP="0")
    assert out["overlapped"] == 0
    check(out, "f64" if case.endswith("f64") else "f32")


@pytest.mark.gpu
def test_rccl_bootstrap_world1():
    """The RCCL communicator (unique-id broadcast through torch.distributed, ncclCommInitRank inside libwlhip)
    comes up and a decomposed run works on it; only world size 1 is possible on a 1-GPU box."""
    out = run_workers("mg_worker.py", 1, "sphere_rccl_f32")
    check(out, "f32")


def _device_count():
    import torch
    return torch.cuda.device_count()        # (does not initialise the GPU in this process)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["sphere_rccl_deep_f32", "sphere_rccl_zper_deep_f32"])
def test_rccl_two_devices(case):
    """The production transport between two REAL devices: ncclSend/ncclRecv halo planes on the comm stream (split
    communicator), ncclAllReduce'd dot products, ncclAllGather hand-over to the replicated coarse levels -- must
    reproduce the undecomposed run exactly like the host-transport twin.  Needs two GPUs: skipped on a 1-GPU box."""
    if _device_count() < 2:
        pytest.skip("needs 2 GPUs (one per rank)")
    out = run_workers("mg_worker.py", 2, case)
    assert out["overlapped"] > 0
    check(out, "f32")


@pytest.mark.gpu
def test_bench_multirank_path():
    """bench.py's N>1 code path (slab set-up, timing reduction, JSON line) with 2 ranks sharing the GPU over gloo."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--comm", "host", "--size", "64",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert "64x64x128" in out["config"]["workload"]
