"""bench.py's launcher logic where no GPU exists: `--gpus N` must refuse to run rather than print a line for fewer devices."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, **env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NSG_BENCH_SINGLE_DEVICE")}
    env.update(env_extra)
    return subprocess.run([sys.executable, "bench.py", *args], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)


def test_gpus_beyond_the_node_are_refused():
    import torch

    want = torch.cuda.device_count() + 2
    p = _run("--gpus", str(want), "--steps", "5")
    assert p.returncode != 0 and f"--gpus {want} needs {want} GPUs" in p.stderr and "n_gpus" not in p.stdout


def test_launcher_and_flag_must_agree():
    p = _run("--gpus", "4", "--steps", "5", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert p.returncode != 0 and "they must agree" in p.stderr and "n_gpus" not in p.stdout


def test_bad_flag():
    assert _run("--gpus", "0").returncode != 0


def test_eight_gpu_line_prices_the_aggregate_against_eight_peaks():
    """C5's line as the driver will read it at N = 8 (no 8-GPU node exists for this build to run on): the `roofline` block is
    the job's 120 B x value against 8 x 8 TB/s, with rank 0's kernel beside it; at N = 1 it is the kernel's own rate."""
    sys.path.insert(0, ROOT)
    import bench

    n, kern_ms = 1 << 20, 0.0228
    value8 = 8 * n / (0.0235e-3)                   # eight ranks, the slowest needing 23.5 us per step
    ach, peak, extra = bench.job_roofline(value8, 8, n, kern_ms, 0.0235)
    assert peak == 8 * 8000.0 and abs(ach - 120 * value8 / 1e9) < 1e-6 and 0.6 < ach / peak < 0.75
    assert abs(extra["rank0_frac_of_one_gpu"] - 120 * n / (kern_ms * 1e-3) / 1e9 / 8000.0) < 1e-12
    assert extra["slowest_rank_avg_launch_us"] == 23.5 and extra["rank0_avg_launch_us"] == 22.8
    ach1, peak1, extra1 = bench.job_roofline(n / (kern_ms * 1e-3), 1, n, kern_ms, kern_ms)
    assert peak1 == 8000.0 and extra1 is None and abs(ach1 - 120 * n / (kern_ms * 1e-3) / 1e9) < 1e-9
