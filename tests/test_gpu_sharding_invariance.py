"""C5's defining property on the hardware a test box has: the job - bench.py's own, BASELINE's C1 / C5 configuration - run over
the SAME global env indices as 1, 2 and 4 ranks (every rank on the box's one GPU, collectives through gloo: the rehearsal mode
bench.py documents) yields bit-identical results: the all-gathered episode returns and EVERY final row of EVERY env (float64
state, observation, t, theta, episode word, last episode length, last step's outputs), whatever the sharding.  Seeds are
`base + global index` and the synthetic actions are counter-based draws per (pool slot, global index)
(`ns_gym_amd.distributed`).  What this cannot cover is RCCL over xGMI itself (one GPU here; a test box also allows at most six
processes on its card, so 8 ranks are rehearsed on the CPU: tests/test_distributed_cpu.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOTAL = 1 << 18


def _run(world, out_dir, port):
    env = dict(os.environ, NSG_BENCH_SINGLE_DEVICE="1", NSG_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    args = ["--gpus", str(world), "--steps", "300", "--warmup", "10", "--envs-per-gpu", str(TOTAL // world), "--no-cpu-baseline",
            "--no-all-configs", "--no-hbm-resident", "--dump-shards", str(out_dir)]
    if world == 1:
        cmd = [sys.executable, "bench.py"] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port), "bench.py"] + args
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def _job_rows(out_dir, world):
    shards = [np.load(os.path.join(out_dir, f"world{world}_rank{r}.npz")) for r in range(world)]
    assert [int(s["lo"]) for s in shards] == [r * (TOTAL // world) for r in range(world)] and int(shards[-1]["hi"]) == TOTAL
    rows = {}
    for k in shards[0].files:
        if k in ("lo", "hi"):
            continue
        axis = 1 if k in ("phys", "theta", "delta_change") else 0      # [F, N] rows concatenate along the env axis
        rows[k] = np.concatenate([s[k] for s in shards], axis=axis)
    return rows, np.load(os.path.join(out_dir, f"world{world}_gathered_returns.npy"))


def test_sharded_job_is_bit_identical_for_1_2_and_4_ranks(tmp_path):
    lines = {w: _run(w, tmp_path, 29580 + w) for w in (1, 2, 4)}
    ref_rows, ref_ret = _job_rows(tmp_path, 1)
    assert ref_ret.shape == (TOTAL,) and (ref_rows["last_length"] > 0).mean() > 0.99     # episodes did finish
    np.testing.assert_array_equal(ref_ret, ref_rows["last_length"].astype(np.float32))   # CartPole: return == length
    for w in (2, 4):
        rows, ret = _job_rows(tmp_path, w)
        assert ret.tobytes() == ref_ret.tobytes(), f"gathered returns differ between 1 and {w} ranks"
        for k, v in ref_rows.items():
            assert rows[k].shape == v.shape and rows[k].tobytes() == v.tobytes(), f"row {k} differs between 1 and {w} ranks"
        d = lines[w]
        assert d["n_gpus"] == w and d["config"]["total_envs"] == TOTAL and d["config"]["gathered_returns"] == TOTAL
        # the N > 1 line prices the AGGREGATE against N x 8 TB/s and carries the gather-inclusive rates
        r = d["roofline"]
        assert r["peak"] == 8000.0 * w and abs(r["achieved"] - 120 * d["value"] / 1e9) / r["achieved"] < 1e-9
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["rank0_avg_launch_us"] > 0
        assert 0 < d["config"]["value_incl_gather"] < d["value"] and d["config"]["value_incl_gather_T1000"] < d["value"] * 1.000001
    assert lines[1]["roofline"]["peak"] == 8000.0 and "rank0_avg_launch_us" not in lines[1]["roofline"]
