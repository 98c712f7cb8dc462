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


def run_group(cmd, env, timeout):
    """run `cmd` in its own process group; on a timeout the WHOLE group is killed (a launcher's worker ranks included: none
    may be left holding the GPU), then the test fails"""
    import signal
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        out, err = p.communicate()
        raise AssertionError(f"timed out after {timeout} s\n" + out[-2000:] + err[-3000:])
    return p.returncode, out, err


def run_workers(script, nproc, *args, timeout=600, allow_fail=False, **extra_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "tests", script), *args]
    rc, out, err = run_group(cmd, env, timeout)
    if allow_fail and rc != 0:
        return {"failed": True, "tail": out[-1500:] + err[-2500:]}
    assert rc == 0, out[-3000:] + err[-3000:]
    lines = [l for l in out.splitlines() if l.startswith("RESULT ")]
    assert lines, out[-2000:] + err[-2000:]
    if len(args) == 1 and "+" in args[0]:          # several cases in one launch (mg_worker.py): {case: result}
        return {r["case"]: r for r in map(lambda l: json.loads(l[7:]), lines)}
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


def test_slab_partition_8_ranks_c4_c5():
    """The 8-way split of BASELINE configs[3] (1024x1024x512) and configs[4] (512^3): SURVEY 8e's table -- local nz per GPU
    64, 32, 16, 8, 4, 2 on levels 1-6, then the children of a coarse plane would live on two ranks -- and which levels the
    default `replicate_cells` = 2^21 keeps as slabs (the rest is replicated after one all-gather per V-cycle)."""
    from waterlily_amd.dist import HZ, Slab, collectives_per_step, plan_levels
    P, nz = 8, 512
    slabs = [Slab(r, P, nz) for r in range(P)]
    owned = []
    for s in slabs:
        assert (s.nzl, s.n2l, s.nzg) == (64, 64 + 2 * HZ, nz + 2)
        owned += [s.kz0 + k for k in range(s.own_lo, s.own_hi + 1)]
    assert owned == list(range(nz + 2))
    ring = [Slab(r, P, nz, ring=True) for r in range(P)]                 # z-periodic: nobody owns the two ghost planes
    assert sum(s.own_hi - s.own_lo + 1 for s in ring) == nz
    s, chain = slabs[5], []
    while s is not None:
        chain.append(s.nzl)
        s = s.coarser()
    assert chain == [64, 32, 16, 8, 4, 2]
    for r in range(P):                                                    # every coarse slab starts on an odd fine plane
        s = slabs[r]
        while s is not None:
            assert (s.kz0 + HZ) % 2 == 1 and (s.kz0 + HZ - 1) == r * s.nzl
            s = s.coarser()
    # "deep": as long as the partition allows -> six slab levels, then 16x16x8 and below replicated
    deep = plan_levels((1026, 1026, 514), slabs[0], replicate_cells=0)
    assert [n for n, _ in deep] == [(1026, 1026, 514), (514, 514, 258), (258, 258, 130), (130, 130, 66), (66, 66, 34), (34, 34, 18),
                                    (18, 18, 10), (10, 10, 6), (6, 6, 4)]
    assert [sl.nzl if sl else None for _, sl in deep] == [64, 32, 16, 8, 4, 2, None, None, None]
    # default: levels of <= 2^21 cells are replicated.  C4: 3 slab levels (128x128x64 = 2^20 cells is the first replicated)
    c4 = plan_levels((1026, 1026, 514), slabs[0])
    assert [sl.nzl if sl else None for _, sl in c4] == [64, 32, 16] + [None] * 6
    # C5 (512^3): 2 slab levels (128^3 = 2^21 cells is replicated)
    c5 = plan_levels((514, 514, 514), slabs[0])
    assert [sl.nzl if sl else None for _, sl in c5] == [64, 32] + [None] * 7
    # undecomposed: same shapes, no slabs (the hierarchy -- hence pois.n -- does not depend on the decomposition)
    assert [n for n, _ in plan_levels((1026, 1026, 514), None)] == [n for n, _ in c4]
    # the per-step collective model at one V-cycle per solve
    assert collectives_per_step(c4, [1, 1]) == {"allreduce": 1 + 2 + 2 * (3 * 13 + 1), "allgather": 2}
    assert collectives_per_step(c5, [1, 1]) == {"allreduce": 1 + 2 + 2 * (2 * 13 + 1), "allgather": 2}
    assert collectives_per_step(plan_levels((1026, 1026, 514), None), [1, 1]) == {"allreduce": 0, "allgather": 0}


@pytest.mark.parametrize("nproc", [2, 8])
def test_host_collectives_gloo(nproc):
    """the host-callback transport over gloo on CPU: all-reduce sum / max, neighbour exchange on a chain and on a periodic
    RING of ranks, in-place all-gather -- at 2 ranks and at the 8 ranks of the BASELINE multi-GPU configurations"""
    out = run_workers("comm_worker.py", nproc, timeout=240)
    assert out["ok"], out


# ----------------------------------------------------------------------------- GPU: decomposed == undecomposed

def check_collectives(out, exitBC=False):
    """the library's own counters for one mom_step! against the model of waterlily_amd.dist.collectives_per_step (the
    figure DESIGN.md section 6 prices a step with): all-reduces and all-gathers exactly, halo traffic as a budget"""
    from waterlily_amd.dist import collectives_per_step
    levels = [(tuple(n), nzl) for (n, _), nzl in zip(out["levels"], out["slab_nzl"])]
    want = collectives_per_step(levels, out["n_counted"], exitBC=exitBC)
    c = out["comm"]
    assert c["allreduce"] == want["allreduce"] and c["allgather"] == want["allgather"], (c, want, out["n_counted"])
    # exchanges: one batch each, never more than one per all-reduce plus the handful of u / f / x exchanges of the step
    assert 0 < c["exchanges"] <= c["allreduce"] + 16, c
    assert c["sendrecv_pairs"] >= c["exchanges"] and c["halo_bytes"] > 0


def check(out, T):
    tol = 2e-5 if T == "f32" else 1e-11
    assert out["n_ref"] == out["n_slab"], (out["n_ref"], out["n_slab"])
    assert np.allclose(out["dt_ref"], out["dt_slab"], rtol=tol)
    for k in ("init_u", "init_mu0", "init_mu1", "init_V"):
        assert out[k] == 0.0, (k, out[k])
    assert out["d_u"] < tol and out["d_f"] < tol and out["d_p"] < 20 * tol, out
    assert np.allclose(out["force_ref"], out["force_slab"], rtol=100 * tol, atol=100 * tol)


SLAB_CASES = [(2, "sphere_deep_f32"), (2, "donut_deep_f64"),
              (2, "sphere_exit_deep_f32"), (2, "sphere_f32"),
              (2, "sphere_zper_deep_f32"), (4, "sphere_long_zper_deep_f64"),
              (2, "sphere_yzper_accel_deep_f64"), (2, "sphere_yper_exit_accel_f32"),
              (2, "sphere_move_deep_f32"), (4, "sphere_long_move_f64"),
              (4, "sphere_vlong_deep_f32"),
              (2, "sphere_oblique_deep_f32"), (4, "sphere_long_oblique_f64")]
_slab_runs = {}


def slab_run(nproc, case):
    """the cases of one rank count run in ONE launch of the ranks (their start-up is most of a small case's wall time); a case
    the launch did not reach -- the worker stops at the first case that raises -- is run on its own, so that its test shows its
    own failure"""
    if nproc not in _slab_runs:
        try:
            _slab_runs[nproc] = run_workers("mg_worker.py", nproc, "+".join(c for n, c in SLAB_CASES if n == nproc), timeout=900)
        except AssertionError as e:
            _slab_runs[nproc] = {"__error__": str(e)[-3000:]}
    got = _slab_runs[nproc].get(case)
    return got if got is not None else run_workers("mg_worker.py", nproc, case)


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,case", SLAB_CASES)
def test_slabs_match_undecomposed(nproc, case):
    out = slab_run(nproc, case)
    # "deep", 32^3 on 2 ranks: levels 32,16,8 (16,8,4 planes per rank) are slabs, 4^3 and 2^3 are replicated;
    # default: only the finest level is a slab (coarser ones hold <= 2^21 cells and are replicated)
    # z-periodic cases ("zper") run on a RING of slabs (rank 0 <-> rank P-1 exchange)
    nslab = sum(1 for _, d in out["levels"] if d)
    assert (nslab >= 3 if "deep" in case else nslab == 1) and not out["levels"][-1][1]
    assert out["overlapped"] > 0          # stencil launches were split around exchanges on the comm stream
    if "vlong" in case:                   # 64x64x128 on 4 ranks: 32, 16, 8, 4, 2 planes per rank, then the hand-over to 4^3
        assert out["slab_nzl"] == [32, 16, 8, 4, 2, None]
    assert out["mailbox"]                 # scalars went through the mailbox all-reduce (the default once a communicator exists)
    if "oblique" in case:                 # a ghost cell of sigma sets the time step: the slabs' shell reductions must find it
        assert out["sigma_max_whole_over_inside"] > 1.5, out["sigma_max_whole_over_inside"]
    check(out, "f64" if case.endswith("f64") else "f32")
    check_collectives(out, exitBC="exit" in case)


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,mailbox", [(2, "1"), (4, "1"), (2, "0")])
def test_allreduce_latency_is_reported(nproc, mailbox):
    """wl_prof_allreduce_us: device-timed cost of one scalar all-reduce as the solver issues them -- through the mailbox (2 and
    4 ranks sharing the GPU) and through the transport's own all-reduce (here: host callbacks over gloo).  Printed with -s;
    the mailbox figure is what DESIGN.md section 6 prices the 8-GPU step with (same PCIe round trips, more peers to poll)."""
    out = run_workers("mg_worker.py", nproc, "arlat_f32", timeout=180, WL_MAILBOX=mailbox)
    print(f"\nall-reduce of one double, {nproc} ranks on one GPU, {'mailbox' if out['mailbox'] else 'transport'}: {out['us_per_allreduce']:.1f} us")
    assert out["mailbox"] == (mailbox == "1") and out["ranks"] == nproc
    assert 0.5 < out["us_per_allreduce"] < (200.0 if out["mailbox"] else 20000.0)


@pytest.mark.gpu
def test_mailbox_allreduce_gives_up_on_a_missing_peer():
    """the waits of the mailbox all-reduce are bounded (wl_set_option(26)): a healthy round sums correctly; when a peer never
    posts, the waiting rank's kernel ends, and the library turns the flag into an error at its next synchronisation"""
    out = run_workers("mg_worker.py", 2, "mboxtimeout_f32", timeout=120)
    assert out["ok_sum"] and out["raised"] and "mailbox" in out["msg"], out


@pytest.mark.gpu
@pytest.mark.parametrize("mailbox", ["1", "0"], ids=["mailbox-scalars", "ncclAllReduce-scalars"])
def test_rccl_two_ranks_on_one_gpu_over_sockets(mailbox):
    """RCCL ITSELF at world size 2 on the one-GPU test box: the two ranks share the device but present themselves to RCCL
    as two hosts (NCCL_HOSTID), so the communicator pairs them over its socket transport.  Everything the library does
    with RCCL runs between two real processes -- ncclCommInitRank from a broadcast unique id, ncclCommSplit for the halo
    communicator, grouped ncclSend/ncclRecv on the comm stream overlapped with split stencil launches, ncclAllReduce per
    dot product, ncclAllGather at the hand-over to the replicated levels -- and must reproduce the undecomposed run.
    What it cannot show is xGMI bandwidth or latency.  If this RCCL build cannot bring up such a communicator at all
    (no usable loopback interface) the test is skipped with RCCL's message; a wrong result or a hang is a failure."""
    out = run_workers("mg_worker.py", 2, "sphere_rcclnet_deep_f32", timeout=300, allow_fail=True, NCCL_DEBUG="WARN", WL_MAILBOX=mailbox)
    if out.get("failed"):
        if "RESULT" not in out["tail"] and ("rccl" in out["tail"].lower() or "nccl" in out["tail"].lower()):
            pytest.skip("RCCL could not create a 2-rank communicator on one GPU: " + out["tail"][-600:])
        raise AssertionError(out["tail"])
    assert out["overlapped"] > 0 and out["mailbox"] == (mailbox == "1")
    check(out, "f32")
    check_collectives(out)


@pytest.mark.gpu
def test_rccl_four_ranks_on_one_gpu_over_sockets():
    """The same with FOUR ranks (32 x 32 x 64, slab levels of 16, 8, 4 and 2 planes per rank): the two middle ranks exchange with a
    neighbour on either side inside one RCCL group, the all-gather collects four segments, the mailbox polls four slots --
    the RCCL call pattern of an interior rank of the 8-GPU runs, between real processes."""
    out = run_workers("mg_worker.py", 4, "sphere_long_rcclnet_deep_f32", timeout=400, allow_fail=True, NCCL_DEBUG="WARN")
    if out.get("failed"):
        if "RESULT" not in out["tail"] and ("rccl" in out["tail"].lower() or "nccl" in out["tail"].lower()):
            pytest.skip("RCCL could not create a 4-rank communicator on one GPU: " + out["tail"][-600:])
        raise AssertionError(out["tail"])
    assert out["overlapped"] > 0 and out["mailbox"] and out["slab_nzl"][:4] == [16, 8, 4, 2]
    check(out, "f32")
    check_collectives(out)


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,case", [(2, "sphere_deep_f32"), (4, "sphere_long_zper_deep_f64")])
def test_slabs_match_without_overlap(nproc, case):
    """Same, with the halo exchanges issued in-stream (WL_OVERLAP=0) instead of on the comm stream with the stencil
    launches split into inner planes (concurrent with the transfer) and the two boundary planes (after it)."""
    out = run_workers("mg_worker.py", nproc, case, WL_OVERLAP="0", WL_MAILBOX="0")
    assert out["overlapped"] == 0 and not out["mailbox"]     # (and the scalars through the transport's own all-reduce)
    check(out, "f64" if case.endswith("f64") else "f32")


@pytest.mark.gpu
def test_two_slabs_at_the_size_of_a_bench_rank():
    """The slab path at the size a rank of `bench.py --gpus N` works on -- 512x512x256 owned cells per rank (the 512^3 sphere on
    two ranks sharing the GPU): the two-rows-per-thread 7-point kernels, the 64x8 conv_diff tiles and the 16 K-workgroup
    streaming kernels in their decomposed form (split launches around the exchanges, partials of three launches laid end
    to end), the mailbox and the hand-over to replicated levels -- must reproduce the undecomposed 512^3 run."""
    out = run_workers("mg_worker.py", 2, "sphere_big_f32", timeout=900)
    assert out["slab_nzl"][:2] == [256, 128] and out["slab_nzl"][-1] is None and out["mailbox"] and out["overlapped"] > 0
    check(out, "f32")
    check_collectives(out)


@pytest.mark.gpu
def test_vtk_write_restart_on_two_slabs(tmp_path):
    """maintests.jl:420-443 on a decomposed run: slab gather on write, slab scatter on restart (2 ranks sharing the GPU)."""
    out = run_workers("mg_worker.py", 2, "vtk_f32", WL_TMP=str(tmp_path))
    assert out["same_u"] and out["same_p"] and out["same_local_u"], out
    assert out["dt"][1] == out["dt"][2] and abs(out["cfl"][0] - out["cfl"][1]) <= 1e-6 * out["cfl"][1]
    assert out["n_next"][0] == out["n_next"][1] and out["d_next"] < 2e-5


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
    """bench.py's N>1 code path under the driver's launcher (slab set-up, timing reduction, JSON line) with 2 ranks sharing the GPU
    over gloo, in its weak-scaling form (--weak: every rank keeps a size^3 slab; the default is strong scaling, covered by
    tests/test_boundary_hosts.py::test_bench_launches_its_own_ranks)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--comm", "host", "--size", "32",
           "--weak", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert "32x32x64" in out["config"]["workload"]
