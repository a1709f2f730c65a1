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
