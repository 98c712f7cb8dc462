"""The drop-in boundary driven by hosts OTHER than the torch-based test harness:

  * tests/ctypes_host.py -- a torch-free process that follows the Julia shim's call sequence (wl_malloc / wl_h2d_2d / pitched or
    dense strides / wl_flow_create / wl_mg_create / wl_mom_step / wl_pforce / wl_d2h_2d / wl_free): the library owns its
    context and allocator;
  * bench.py --gpus N without a launcher around it: the parent starts its ranks as child processes, never touches the GPU,
    relays rank 0's JSON line and the children's status."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "ctypes_host.py")
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


# ----------------------------------------------------------------------------- CPU: host logic, failure paths

def test_ctypes_host_needs_no_torch_and_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, HOST, "sim_2d_f64"], capture_output=True, text=True, env=_env(), timeout=120)
    assert r.returncode != 0
    assert "wl_device_count failed" in (r.stdout + r.stderr) or "no GPU" in (r.stdout + r.stderr)


def test_bench_self_launch_command_and_defaults():
    """the launcher line of the bench contract, and the N>1 default workload (BASELINE configs[3], strong scaling)"""
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.self_launch_cmd(8, 29511, ["--gpus", "8", "--steps", "3"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "8", "--steps", "3"]
    assert cmd[cmd.index("--master-port") + 1] == "29511" and os.path.samefile(cmd[-5], BENCH)
    a = bench.parse_args(["--gpus", "8"])
    assert a.size is None and a.grid is None and not a.weak and bench.C4_GRID == (1024, 1024, 512)


def test_bench_parent_stays_off_the_gpu_and_relays_the_ranks_status():
    """Without a GPU the two ranks fail; the parent (which must not import torch before it decides to launch) reports
    their failure with a non-zero exit and prints no result line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2', '--comm', 'host', '--size', '32', '--steps', '1', '--warmup', '0'];\n"
            "import bench\n"
            "try:\n    bench.main(sys.argv[1:])\nexcept SystemExit as e:\n"
            "    print('PARENT_TORCH', 'torch' in sys.modules, 'RC', e.code); raise\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=_env(), cwd=ROOT, timeout=600)
    assert r.returncode != 0
    assert "PARENT_TORCH False" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert '{"metric"' not in r.stdout


# ----------------------------------------------------------------------------- GPU

@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["pitched", "dense"])
@pytest.mark.parametrize("name", ["sim_3d_f32", "sim_2d_f64"])
def test_torch_free_host_reproduces_the_golden_steps(name, layout):
    """ext/WaterLilyAMDGPUExt.jl's role played through the C ABI alone: memory from wl_malloc, handles, steps, force, read-back
    -- against tests/golden (u, p, pois.n, Δt, pressure force).  pitched: the shim's HIPArray (rows on 128-byte boundaries,
    wl_h2d_2d / wl_d2h_2d); dense: the reference's own strides."""
    r = subprocess.run([sys.executable, HOST, name, layout], capture_output=True, text=True, env=_env(), timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["ok"] and not res["torch_imported"] and res["n"] == res["n_expected"], res


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the two ranks share the test GPU over the host transport): one
    JSON line, two ranks, strong scaling on the stated grid, the one-GPU time of the same grid and the speed-up in the line."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--comm", "host", "--size", "64", "--steps", "2", "--warmup", "1",
                        "--ref1-steps", "2"], capture_output=True, text=True, env=_env(), cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["comm_ranks"] == 2 and out["scaling"] == "strong"
    assert "64x64x64" in out["config"]["workload"] and out["config"]["transport"].startswith("host")
    assert out["one_gpu"]["ms_per_step"] > 0 and out["speedup_vs_1gpu"] > 0
    assert out["value"] > 0 and out["config"]["collectives_last_step"]["allreduce"] > 0


@pytest.mark.gpu
def test_bench_rccl_path_between_two_processes():
    """bench.py's PRODUCTION transport path (--comm rccl: ncclCommInitRank from a broadcast id, the mailbox, RCCL halo exchanges and
    all-gathers, the communicator assertion, rank 0's one-GPU leg after the communicator is gone) between two real processes on the
    one GPU of the box: they present themselves to RCCL as two hosts, so it pairs them over sockets (WL_RCCL_OVER_SOCKETS=1)."""
    env = dict(_env(), WL_RCCL_OVER_SOCKETS="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--size", "64", "--steps", "2", "--warmup", "1", "--ref1-steps", "2"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    if r.returncode != 0 and "RCCL communicator failed" in r.stderr and '{"metric"' not in r.stdout:
        pytest.skip("RCCL could not pair two ranks on one GPU over sockets on this box: " + r.stderr[-600:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    assert out["n_gpus"] == 2 and out["config"]["comm_ranks"] == 2 and out["config"]["transport"].startswith("rccl")
    assert out["config"]["scalar_allreduce"].startswith("mailbox") and out["speedup_vs_1gpu"] > 0
    cl = out["config"]["collectives_last_step"]
    assert cl["allreduce"] > 0 and cl["exchanges"] > 0 and max(out["config"]["vcycles_per_solve"]) <= 3


@pytest.mark.gpu
def test_bench_loopback_rank_runs_the_slab_of_one_rank():
    """bench.py --comm loopback: one process plays a rank of an N-way split -- its slab with halos, split launches, reductions
    and (device-copy) exchanges, the body shrunk to fit the slab.  Where the solver converges (two ranks at BASELINE size) the
    line carries the per-rank step time; where the copy-of-itself neighbours do not fit the replicated coarse levels the solves
    stall, and the line SAYS so instead of quoting a step time (only its per-launch class times are usable)."""
    def run(n):
        r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--comm", "loopback", "--grid", "64", "64", "128",
                            "--steps", "2", "--warmup", "4", "--no-cpu-baseline"], capture_output=True, text=True, env=_env(), cwd=ROOT,
                           timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    out = run(2)
    assert out["loopback"]["rank"] == 1 and out["loopback"]["of"] == 2 and out["config"]["comm_ranks"] == 2
    cl = out["config"]["collectives_last_step"]
    assert cl["allreduce"] > 0 and cl["exchanges"] > 0
    out8 = run(8)
    assert out8["loopback"]["of"] == 8 and out8["config"]["collectives_last_step"]["allgather"] > 0
    for o in (out, out8):      # (at BASELINE size the 2-way rank converges; on this small grid it may not)
        lb = o["loopback"]
        if lb["solver_converged"]:
            assert lb["per_rank_ms_per_step"] > 0 and lb["caveat"] is None and max(o["config"]["vcycles_per_solve"]) < 32
        else:
            assert lb["per_rank_ms_per_step"] is None and "STALLED" in lb["caveat"]
