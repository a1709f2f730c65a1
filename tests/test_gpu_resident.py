"""The resident stepper (nsg_resident_start; VecNSEnv: ResidentStepper): one launch that stays on the device and steps whenever
the producer publishes the next action row.  Bit-identical to the same number of nsg_step calls - resets, fires and all -, every
wait bounded (a silent producer costs the budget + grace period, never a hung process), `stop` honoured, batches beyond one
workgroup per chunk refused."""
import time

import numpy as np
import pytest
import torch

from ns_gym_amd import workloads as W
from ns_gym_amd.vec_env import ResidentStepper

pytestmark = pytest.mark.gpu

ROWS = ("phys", "cell", "theta", "table_prob", "t", "status", "episode", "rng_env", "rng_upd", "cursor", "obs", "reward", "terminated",
        "truncated", "env_change", "delta_change", "prob")


def _policy(env, k, out):
    """The library's demo policy as torch ops: ((obs[:, 2] > 0) + k) mod n_actions."""
    torch.remainder((env.state[:, 2] > 0).to(torch.int32) + k, env.n_actions, out=out)


def _same(a, b, what):
    for r in ROWS:
        if a.buf[r] is not None:
            assert torch.equal(a.buf[r], b.buf[r]), f"{what}: row {r} differs"
    ca, cb = a.counters(), b.counters()
    assert ca == cb, (what, ca, cb)


@pytest.mark.parametrize("name,n,specialize", [("c2", 1 << 16, True), ("c2", 5000, False), ("c1", 30000, True), ("acro", 4096, False)])
def test_closed_loop_is_bit_identical_to_step_calls(name, n, specialize):
    K = 700 if name != "acro" else 520      # (Acrobot: past its TimeLimit of 500, so every env is reset inside the loop)
    ref = W.build(name, n, specialize=specialize, seed=11, track_returns=False)
    env = W.build(name, n, specialize=specialize, seed=11, track_returns=False)
    a = torch.zeros(n, dtype=torch.int32, device="cuda")
    for k in range(K):
        _policy(ref, k, a)
        ref.step(a)
    assert ref.counters()["episodes"] >= n and ref.counters()["updates_applied"] > 0       # resets and fires happened inside the loop
    loop = ResidentStepper(env, torch.zeros(n, dtype=torch.int32, device="cuda"), wait_budget_us=50_000)
    loop.start(K)
    loop.demo_policy(K, stream=torch.cuda.Stream())
    status, steps = loop.result()
    assert (status, steps) == ("finished", K)
    _same(env, ref, f"{name} after {K} resident steps")
    # ordinary calls carry on from where the resident kernel left the batch
    for k in range(K, K + 5):
        _policy(ref, k, a); ref.step(a)
        _policy(env, k, a); env.step(a)
    _same(env, ref, "after 5 more ordinary steps")
    env.close(); ref.close()


def test_a_silent_producer_costs_the_budget_not_the_process():
    env = W.build("c2", 1 << 16, specialize=False, seed=1, track_returns=False)
    before = {r: env.buf[r].clone() for r in ROWS if env.buf[r] is not None}
    loop = ResidentStepper(env, torch.zeros(env.N, dtype=torch.int32, device="cuda"), wait_budget_us=3000)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop.start(1000)                      # nobody ever publishes an action row
    status, steps = loop.result()
    ms = (time.perf_counter() - t0) * 1e3
    assert (status, steps) == ("starved", 0)
    assert 3.0 <= ms < 30.0, ms           # budget 3 ms + grace 0.2 ms (+ launch, + the first call's clock calibration)
    for r, v in before.items():
        assert torch.equal(env.buf[r], v), r          # untouched
    env.step(torch.zeros(env.N, dtype=torch.int32, device="cuda"))      # and usable
    env.close()


def test_host_driven_rows_then_silence_leaves_exactly_those_steps():
    """A producer that publishes three rows (here: torch ops on a side stream, per-chunk words filled at once) and then goes
    silent: the kernel takes exactly three steps everywhere, says `starved`, and the batch equals three nsg_step calls."""
    n = 20000
    env = W.build("c3", n, specialize=False, seed=2, track_returns=False)      # a grid env: cell / prob rows are written through every step
    ref = W.build("c3", n, specialize=False, seed=2, track_returns=False)
    acts = [torch.randint(0, 4, (n,), dtype=torch.int32, device="cuda") for _ in range(3)]
    for a in acts:
        ref.step(a)
    buf = torch.zeros(n, dtype=torch.int32, device="cuda")
    # (the budget is what the kernel waits for a row before it calls itself starved: generous, so that a cold process - first use of
    # torch's copy / reduction kernels on the producer's side - cannot lose the first row; the ops are warmed up anyway)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        buf.copy_(acts[0]); int(buf.to(torch.int64).min()); buf.zero_()
    side.synchronize()
    loop = ResidentStepper(env, buf, wait_budget_us=300_000)
    loop.start(100)
    side.wait_event(loop._zeroed)
    with torch.cuda.stream(side):
        for k, a in enumerate(acts):
            t0 = time.perf_counter()
            while int(loop.step_seq.min()) < k:      # (host round trips: slow, but this is the protocol)
                assert time.perf_counter() - t0 < 5.0, f"step {k - 1} was never published (step_seq = {loop.step_seq.tolist()[:8]} ...)"
            buf.copy_(a)
            loop.publish(k, stream=side)             # behind the copy on ITS stream: act_seq[j] = k + 1 for every chunk
    status, steps = loop.result()
    assert (status, steps) == ("starved", 3) and loop.steps_range == (3, 3)     # a publish-for-all producer: the chunks agree
    _same(env, ref, "three host-driven resident steps")
    env.close(); ref.close()


def test_stop_ends_the_loop_and_every_chunk_is_where_its_own_count_says():
    """`stop` raised from outside: producer and stepper leave within the grace period.  The demo policy's workgroups run
    independently of each other, so the chunks stand at their own step counts (step_seq[j]); every chunk's rows are the
    reference's after exactly that many steps."""
    n, K = 1 << 14, 10_000_000
    env = W.build("c2", n, specialize=True, seed=5, track_returns=False)
    loop = ResidentStepper(env, torch.zeros(n, dtype=torch.int32, device="cuda"), wait_budget_us=50_000)
    loop.start(K)
    loop.demo_policy(K, stream=torch.cuda.Stream())
    time.sleep(0.01)
    loop.stop()
    status, lo = loop.result()
    lo, hi = loop.steps_range
    counts = loop.step_seq.cpu().numpy()
    assert status == "stopped" and 100 < lo <= hi < K and counts.min() == lo and counts.max() == hi and hi - lo < 2000, (status, lo, hi)
    ref = W.build("c2", n, specialize=True, seed=5, track_returns=False)
    a = torch.zeros(n, dtype=torch.int32, device="cuda")
    snaps = {}
    for k in range(hi):
        _policy(ref, k, a); ref.step(a)
        if k + 1 >= lo:
            snaps[k + 1] = (ref.state.clone(), ref.t.clone(), ref.theta.clone(), ref.buf["rng_upd"].clone())
    for j, c in enumerate(counts):
        sl = slice(256 * j, 256 * (j + 1))
        obs, t, theta, rng = snaps[int(c)]
        assert torch.equal(env.state[sl], obs[sl]) and torch.equal(env.t[sl], t[sl]) and torch.equal(env.theta[:, sl], theta[:, sl]), f"chunk {j} at {c} steps"
        assert torch.equal(env.buf["rng_upd"].view(-1, n, 4)[:, sl], rng.view(-1, n, 4)[:, sl]), f"chunk {j}: update-fn streams"
    env.close(); ref.close()


def test_batches_beyond_one_workgroup_per_chunk_are_refused():
    from ns_gym_amd._lib import NsgError

    env = W.build("c1", (1 << 17) + 256, specialize=False, track_returns=False)
    loop = ResidentStepper(env, torch.zeros(env.N, dtype=torch.int32, device="cuda"))
    with pytest.raises(NsgError, match="at most 131072 envs"):
        loop.start(10)
    env.close()
