"""bench.py's output contract, exercised the way the driver calls it: one JSON line on stdout with the agreed keys - as a
single process, and as two ranks under torch.distributed.run (both on the one GPU of a test box, collectives through
gloo: the rehearsal mode bench.py documents; the driver's real runs use one GPU per rank and RCCL)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline"}


def _line(out):
    lines = [ln for ln in out.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_process_line():
    p = subprocess.run([sys.executable, "bench.py", "--steps", "60", "--warmup", "10", "--envs-per-gpu", "65536", "--cpu-seconds", "0.5"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    assert KEYS <= set(d) and "cpu_baseline" in d
    assert d["n_gpus"] == 1 and d["steps"] == 60 and d["warmup"] == 10 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "env-steps/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert abs(d["value"] - 65536 * 60 / (d["ms_per_step"] * 60 / 1e3)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - 120 * 65536 / (r["avg_launch_us"] * 1e-6) / 1e9) / r["achieved"] < 1e-6
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and "workload" in d["config"]
    assert d["cpu_baseline"]["numpy_vectorised"]["value"] > 0 and d["cpu_baseline"]["numpy_vectorised"]["cores"] == 1
    assert abs(r["frac_from_ms_per_step"] - 120 * d["value"] / 8e12) < 1e-9 and d["config"]["gather_verified"] is True
    assert "static" in r["traffic_source"] and "Infinity Cache" in r["residency"]
    h = d["roofline_hbm_resident"]                      # the same kernel on 2^24 envs: rows stream from HBM
    assert h["envs"] == 1 << 24 and h["bound"] == "hbm" and abs(h["frac"] - h["achieved"] / 8000.0) < 1e-9
    assert h["launches"] == 100 and h["repetitions"] == 3 and len(h["repetitions_us"]) == 3
    assert abs(h["achieved"] - 120 * (1 << 24) / (h["avg_launch_us"] * 1e-6) / 1e9) / h["achieved"] < 1e-6
    # every BASELINE configuration at the size BASELINE quotes it at, each priced with its own bytes
    own = d["config"]["baseline_configs_at_own_size"]
    assert own["C2"]["envs"] == 65536 and own["C3"]["envs"] == 1 << 20 and own["C4"]["envs"] == [1 << 18, 1 << 18]
    assert "specialised" in own["C4"]["launch"] and own["C4"]["rollout_k64_env_steps_per_sec"] > 0
    for tag, by in (("C2", 157), ("C3", 96)):
        r = own[tag]
        assert abs(r["frac_of_hbm_peak"] - by * r["envs"] / (r["step_us"] * 1e-6) / 8e12) < 1e-9 and r["rollout_k64_env_steps_per_sec"] > 0
    assert abs(own["C4"]["frac_of_hbm_peak"] - 210 * (1 << 18) / (own["C4"]["step_us"] * 1e-6) / 8e12) < 1e-9
    # a single-GPU line carries the gather-inclusive rates too (the "gather" is the local read-out there)
    assert 0 < d["config"]["value_incl_gather"] <= d["value"]


def test_two_ranks_aggregate_line():
    env = dict(os.environ, NSG_BENCH_SINGLE_DEVICE="1", NSG_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29571", "bench.py", "--gpus", "2", "--steps", "60", "--warmup", "10", "--envs-per-gpu", "65536"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    assert KEYS <= set(d) and "cpu_baseline" not in d            # the CPU leg runs on rank 0 of an N = 1 run only
    assert d["n_gpus"] == 2 and d["config"]["total_envs"] == 2 * 65536 and d["config"]["gathered_returns"] == 2 * 65536
    assert abs(d["value"] - 2 * 65536 * 60 / (d["ms_per_step"] * 60 / 1e3)) / d["value"] < 1e-6   # whole-job aggregate
    assert d["config"]["returns_gather_ms"] > 0.0       # the end-of-rollout exchange, timed on its own
    # first-contact diagnostics of an N > 1 run: the gathered CONTENT was checked, both gather schedules were timed, and a rank
    # whose host loop or kernel lags shows in the line
    c = d["config"]
    assert c["gather_verified"] is True and c["returns_gather_schedule"] == "rccl"
    assert c["returns_gather_ms_rccl_allgather"] > 0.0 and c["returns_gather_ms_direct_p2p"] > 0.0
    assert c["slowest_rank_host_loop_steps_per_s"] <= c["rank0_host_loop_steps_per_s"] * 1.0000001 and c["slowest_rank_host_loop_steps_per_s"] > 0
    assert c["slowest_rank_avg_launch_us"] >= c["rank0_avg_launch_us"] * 0.9999999 > 0
    assert abs(d["roofline"]["frac_from_ms_per_step"] - 120 * d["value"] / 1e9 / 16000.0) < 1e-9


def test_single_rank_through_rccl():
    """NSG_BENCH_FORCE_DIST=1: one rank opens an RCCL (backend "nccl") process group anyway, so the barriers, the all-gather of
    episode returns and the max-over-ranks reduction of the N > 1 path execute through the library a one-GPU box can otherwise
    never reach (two ranks cannot share a device under RCCL)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NSG_BENCH_SINGLE_DEVICE", "NSG_BENCH_BACKEND")}
    import socket

    with socket.socket() as sock:                      # a free rendezvous port (the launcher-driven tests use fixed ones)
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env.update(NSG_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    p = subprocess.run([sys.executable, "bench.py", "--steps", "40", "--warmup", "10", "--envs-per-gpu", "65536", "--no-cpu-baseline",
                        "--no-all-configs", "--no-hbm-resident"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    assert d["n_gpus"] == 1 and d["config"]["collectives"] == "torch.distributed backend nccl, 1 rank(s)"
    assert d["config"]["gathered_returns"] == 65536 and d["config"]["returns_gather_ms"] > 0.0 and d["config"]["gather_verified"] is True


def test_gpus_flag_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (here both on the one GPU of the test box,
    collectives through gloo) and prints ONE line with n_gpus = 2."""
    env = dict(os.environ, NSG_BENCH_SINGLE_DEVICE="1", NSG_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "40", "--warmup", "10", "--envs-per-gpu", "65536"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    assert d["n_gpus"] == 2 and d["config"]["total_envs"] == 2 * 65536 and d["config"]["gathered_returns"] == 2 * 65536


def test_more_ranks_than_gpus_is_refused():
    """A one-GPU box must not answer `--gpus 8` with an N = 1 line."""
    import torch

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NSG_BENCH_SINGLE_DEVICE")}
    want = torch.cuda.device_count() + 7
    p = subprocess.run([sys.executable, "bench.py", "--gpus", str(want), "--steps", "5"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0 and "n_gpus" not in p.stdout and f"--gpus {want} needs {want} GPUs" in p.stderr
    # and a launcher that started a different number of ranks than --gpus says is refused as well
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "5"], cwd=ROOT, env=env2, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode != 0 and "they must agree" in p.stderr and "n_gpus" not in p.stdout
