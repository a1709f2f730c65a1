#!/usr/bin/env python3
"""bench.py — non-stationary env-steps/sec of the fused HIP stepper (BASELINE.json metric).

A "step" is one pass of the hot path (one nsg_step launch) over one batch of N envs with
synthetic random actions already resident in HBM.  Workload at every N_gpus: the C1/C5
configuration of BASELINE.json — CartPole-v1 + masspole IncrementUpdate(+0.1) via
ContinuousScheduler — at 2^20 envs PER GPU (weak scaling; 8 GPUs = C5's 8,388,608 envs).
Multi-GPU: one process per GPU (torch.distributed, backend nccl == RCCL); envs are sharded by
contiguous index with no per-step collective; the only exchange is one all-gather of the
per-env episode returns at rollout END - per rollout, not per step - so it is timed on its own,
right after the K timed steps and between the same barriers, OUTSIDE the K-step region that
`value` is computed from (`config.returns_gather_ms`; `config.value_incl_gather` is the rate with
the gather's time added to the K steps', `config.value_incl_gather_T1000` the same for a
1000-step rollout).  Seeds AND actions are functions of the GLOBAL env index, so every env walks
the same trajectory however the job is sharded (tests/test_gpu_sharding_invariance.py).

`python bench.py --gpus N` without a torch.distributed launcher around it starts the N ranks ITSELF (one child per GPU under
torch.distributed.run, before this process touches a GPU) and relays rank 0's line; under a launcher (WORLD_SIZE set) it is
one rank and `--gpus` must equal WORLD_SIZE.  It refuses to run on fewer devices than ranks.

Prints ONE JSON line on rank 0 (see the task contract) with `roofline`, `roofline_hbm_resident` and `cpu_baseline`.
At N > 1 `roofline` prices the AGGREGATE: 120 B x `value` against N x 8 TB/s, rank 0's own kernel time beside it.
`config.baseline_configs_at_own_size`: every BASELINE configuration at the batch size BASELINE quotes it at (C1 2^20, C2 65 536,
C3 2^20, C4 Pendulum 2^18 + Acrobot 2^18 in one heterogeneous launch), step API and fused rollout, each with its own bytes; its
`libm_exact_step_us` times C1 / C2 / C4 through the opt-in bit-exact units (information, never `value`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PER_GPU = 1 << 20
N_HBM_RESIDENT = 1 << 24       # second roofline figure: a batch whose rows (2.5 GB) cannot live in the 256-MiB Infinity Cache
BYTES_PER_ENV_STEP = 120       # C1/C5, fp64 internal state
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(n_envs, steps, threads):
    """The oracle (C port of the reference path) on the host cores: bounded sample of the same
    workload.  Test infrastructure used only as the reported CPU baseline."""
    import numpy as np

    from ns_gym_amd import make
    from ns_gym_amd.schedulers import ContinuousScheduler
    from ns_gym_amd.update_functions import IncrementUpdate
    from oracle.oracle import OracleVecEnv

    env = OracleVecEnv(make("CartPole-v1"), {"masspole": IncrementUpdate(ContinuousScheduler(), k=0.1)}, n_envs,
                       change_notification=True, delta_change_notification=True, track_returns=True)
    env.reset(seed=0)
    acts = np.random.default_rng(123).integers(2, size=(8, n_envs)).astype(np.int32)
    for k in range(3):
        env.step_mt(acts[k % 8], threads)
    t0 = time.perf_counter()
    for k in range(steps):
        env.step_mt(acts[k % 8], threads)
    dt = time.perf_counter() - t0
    return n_envs * steps / dt, dt


def baseline_configs_at_own_size(dev, c1_step_us, c1_rollout_rate, n_c1):
    """Every BASELINE configuration at the batch size BASELINE quotes it at, on this GPU, config-specialised kernels: the step
    API (average of 300 back-to-back launches, HIP events on the launch stream) against ITS algorithmic bytes and the 8 TB/s
    peak, and the fused rollout (K = 64) with the bytes a rollout moves (action + recorded outputs).  C1's row repeats the
    line's own measurement.  Not `value`."""
    import torch

    from ns_gym_amd import workloads as W
    from ns_gym_amd.vec_env import step_group, step_group_kind

    def frac(bytes_per_step, n_envs, us):
        return bytes_per_step * n_envs / (us * 1e-6) / 1e9 / HBM_PEAK_GBS

    def rollout_us(env, K=64, reps=4):
        acts = torch.stack([W.random_actions(env) for _ in range(K)])
        env.rollout(acts)
        torch.cuda.synchronize()
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record()
        for _ in range(reps):
            env.rollout(acts)
        r1.record()
        torch.cuda.synchronize()
        return r0.elapsed_time(r1) * 1e3 / (reps * K)

    rows = {"C1": {"envs": n_c1, "step_us": c1_step_us, "bytes_per_env_step": BYTES_PER_ENV_STEP,
                   "frac_of_hbm_peak": frac(BYTES_PER_ENV_STEP, n_c1, c1_step_us), "env_steps_per_sec": n_c1 / (c1_step_us * 1e-6),
                   "rollout_k64_env_steps_per_sec": c1_rollout_rate}}
    for tag, name in (("C2", "c2"), ("C3", "c3")):
        w = W.WORKLOADS[name]
        env = W.build(name, specialize=True, device=dev)
        a = W.random_actions(env)
        for _ in range(30):
            env.step(a)
        us = min(env.time_steps(a, 300) for _ in range(2)) * 1e3
        ru = rollout_us(env)
        rows[tag] = {"envs": env.N, "kernels": env.kernels, "step_us": us, "bytes_per_env_step": w["bytes_per_env_step"],
                     "frac_of_hbm_peak": frac(w["bytes_per_env_step"], env.N, us), "env_steps_per_sec": env.N / (us * 1e-6),
                     "rollout_k64_us_per_step": ru, "rollout_k64_env_steps_per_sec": env.N / (ru * 1e-6)}
        try:   # the same batch as a CLOSED loop whose policy is evaluated inside the kernel (nsg_rollout_policy, K = 64 per launch)
            rows[tag].update(_fused_closed_loop_us(env, name))
        except Exception as e:
            rows[tag]["fused_closed_loop_error"] = f"{type(e).__name__}: {e}"[:200]
        if env.N <= 1 << 16:
            rows[tag]["regime"] = "one wavefront per SIMD: bound by launch latency + the step's serial chain, not by memory"
            try:   # the same batch as a CLOSED loop without a launch per step (nsg_resident_start + the resident demo policy)
                rows[tag].update(_resident_loop_us(env, a))
            except Exception as e:
                rows[tag]["resident_error"] = f"{type(e).__name__}: {e}"[:200]
        env.close()
    pend, acro = W.build("pend", specialize=True, device=dev), W.build("acro", specialize=True, device=dev)
    ap_, aa = W.random_actions(pend), W.random_actions(acro)
    for _ in range(20):
        step_group([pend, acro], [ap_, aa])
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300):
            step_group([pend, acro], [ap_, aa])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 300)
    # ... and the pair through ONE fused launch of K = 64 steps (nsg_rollout_group)
    from ns_gym_amd.vec_env import rollout_group

    K = 64
    gacts = [torch.stack([W.random_actions(e) for _ in range(K)]) for e in (pend, acro)]
    for _ in range(3):
        rollout_group([pend, acro], gacts)
    torch.cuda.synchronize()
    r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    r0.record()
    for _ in range(8):
        rollout_group([pend, acro], gacts)
    r1.record()
    torch.cuda.synchronize()
    group_rollout_us = r0.elapsed_time(r1) * 1e3 / (8 * K)
    by = W.WORKLOADS["pend"]["bytes_per_env_step"] + W.WORKLOADS["acro"]["bytes_per_env_step"]
    fused = {}
    try:   # C4 as CLOSED loops with in-kernel policies: one nsg_rollout_policy launch per member per 64 steps
        fp, fa = _fused_closed_loop_us(pend, "pend"), _fused_closed_loop_us(acro, "acro")
        fused = {"fused_closed_loop_us_per_step": fp["fused_closed_loop_us_per_step"] + fa["fused_closed_loop_us_per_step"],
                 "fused_closed_loop_members_us_per_step": [fp["fused_closed_loop_us_per_step"], fa["fused_closed_loop_us_per_step"]],
                 "fused_closed_loop_policy": "a linear policy per member, evaluated in the kernel; two launches (one per member) per 64 steps"}
    except Exception as e:
        fused = {"fused_closed_loop_error": f"{type(e).__name__}: {e}"[:200]}
    rows["C4"] = {**fused, "envs": [pend.N, acro.N], "launch": "one nsg_step_group launch (" + step_group_kind([pend, acro]) + " unit)",
                  "step_us": best, "bytes_per_env_pair_step": by, "frac_of_hbm_peak": frac(by, pend.N, best),
                  "env_steps_per_sec": (pend.N + acro.N) / (best * 1e-6),
                  "rollout_k64_us_per_step": group_rollout_us, "rollout_k64_env_steps_per_sec": (pend.N + acro.N) / (group_rollout_us * 1e-6),
                  "pendulum_alone_us": min(pend.time_steps(ap_, 200) for _ in range(2)) * 1e3,
                  "acrobot_alone_us": min(acro.time_steps(aa, 200) for _ in range(2)) * 1e3,
                  "bound": "the Acrobot member's float64 vector-ALU issue (RK4: 15 sincos + 12 divisions per step), not HBM"}
    pend.close(); acro.close()
    try:   # what the opt-in bit-exact arithmetic (NSG_F_LIBM_EXACT: libm's sin / cos / pow / exp, rounding for rounding) costs per step
        rows["libm_exact_step_us"] = _libm_exact_step_us(dev, n_c1)
    except Exception as e:
        rows["libm_exact_error"] = f"{type(e).__name__}: {e}"[:200]
    return rows


def _libm_exact_step_us(dev, n_c1):
    """`VecNSEnv(libm_exact=True)`: float64 state equal to the reference's in every bit (tests/test_gpu_libm_exact.py).  The same
    configs at the same sizes as the rows above, through the exact unit (hiprtc builds it here); C4's pair in its one group launch."""
    import torch

    from ns_gym_amd import workloads as W
    from ns_gym_amd.vec_env import step_group

    out = {}
    for tag, name, n in (("C1", "c1", n_c1), ("C2", "c2", None)):
        env = W.build(name, n, specialize=True, device=dev, libm_exact=True)
        a = W.random_actions(env)
        for _ in range(30):
            env.step(a)
        out[tag] = min(env.time_steps(a, 300) for _ in range(2)) * 1e3
        env.close()
    pend, acro = W.build("pend", specialize=True, device=dev, libm_exact=True), W.build("acro", specialize=True, device=dev, libm_exact=True)
    ap_, aa = W.random_actions(pend), W.random_actions(acro)
    for _ in range(20):
        step_group([pend, acro], [ap_, aa])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        step_group([pend, acro], [ap_, aa])
    e1.record()
    torch.cuda.synchronize()
    out["C4"] = e0.elapsed_time(e1) * 1e3 / 200
    pend.close(); acro.close()
    return out


def _fused_closed_loop_us(env, name, K=64, reps=8):
    """us per step of a closed loop that never leaves the kernel: a linear policy on the observation (classic control) or a table over
    the cells (grid envs) decides inside nsg_rollout_policy, the episode accounts stay in registers; K steps per launch."""
    import numpy as np
    import torch

    from ns_gym_amd.policies import EpisodeAccounts, LinearPolicy, TabularPolicy

    if env.is_grid:
        pol, what = TabularPolicy(np.random.default_rng(0).integers(0, env.n_actions, size=env.cfg.nrow * env.cfg.ncol)), "a table over the cells"
    elif env.obs_dim == 4 and env.n_actions == 2:
        pol, what = LinearPolicy(np.array([[0.3, -0.8, -2.0, -1.1, 0.05], [-0.3, 0.8, 2.0, 1.1, -0.05]], dtype=np.float32)), "a linear policy on the observation"
    else:   # any other classic-control env type: fixed pseudo-random weights of the right shape
        rows = 1 if env.action_is_float else env.n_actions
        pol, what = LinearPolicy(np.random.default_rng(0).normal(size=(rows, env.obs_dim + 1)).astype(np.float32)), "a linear policy on the observation"
    acc = EpisodeAccounts(env, gamma=None)
    for _ in range(3):
        env.rollout_policy(pol, K, accounts=acc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        env.rollout_policy(pol, K, accounts=acc)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * K)
    return {"fused_closed_loop_us_per_step": us, "fused_closed_loop_env_steps_per_sec": env.N / (us * 1e-6),
            "fused_closed_loop_policy": f"{what}, evaluated in the kernel ({env.policy_kernels} unit); {K} steps per launch"}


def _resident_loop_us(env, a, K=4000):
    """us per step of ONE resident launch taking K steps: closed loop (the library's resident demo policy on a second stream reads
    each step's observation and publishes the next action row, per 256-env chunk) and open loop (rows published in advance)."""
    import torch

    from ns_gym_amd.vec_env import ResidentStepper

    loop = ResidentStepper(env, a.clone(), wait_budget_us=50_000)
    pol = torch.cuda.Stream()
    out = {}
    for mode in ("closed", "open"):
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record(loop.stream)
            loop.start(K, prefilled=K if mode == "open" else 0)
            if mode == "closed":
                loop.demo_policy(K, stream=pol)
            e1.record(loop.stream)
            status, steps = loop.result()
            if (status, steps) != ("finished", K):
                raise RuntimeError(f"resident loop ended {status} after {steps} steps")
            if rep:
                best = min(best, e0.elapsed_time(e1) * 1e3 / K)
        out[f"resident_{mode}_loop_us_per_step"] = best
    out["resident_closed_loop_env_steps_per_sec"] = env.N / (out["resident_closed_loop_us_per_step"] * 1e-6)
    return out


def job_roofline(value, world, n_per_gpu, kern_ms_rank0, kern_ms_slowest):
    """(achieved GB/s, peak GB/s, extra fields) of the line's `roofline` block.  One GPU: the kernel's own rate - algorithmic bytes
    per launch over its average launch duration - against 8 TB/s.  N > 1: the job's aggregate - 120 B x all ranks' env-steps per
    second of the job's (max-over-ranks) time - against N x 8 TB/s, rank 0's own kernel beside it."""
    achieved = BYTES_PER_ENV_STEP * n_per_gpu / (kern_ms_rank0 * 1e-3) / 1e9
    if world == 1:
        return achieved, HBM_PEAK_GBS, None
    multi = {"rank0_avg_launch_us": kern_ms_rank0 * 1e3, "rank0_frac_of_one_gpu": achieved / HBM_PEAK_GBS,
             "slowest_rank_avg_launch_us": kern_ms_slowest * 1e3}
    return BYTES_PER_ENV_STEP * value / 1e9, HBM_PEAK_GBS * world, multi


def self_launch(args, argv):
    """`--gpus N > 1` outside a launcher: run N ranks under torch.distributed.run as a child (this process never initialises a
    GPU: device_count() does not), relay its output, exit with its code."""
    import socket
    import subprocess

    import torch

    single = os.environ.get("NSG_BENCH_SINGLE_DEVICE") == "1"    # rehearsal: every rank on cuda:0 (tests on a one-GPU box)
    have = torch.cuda.device_count()
    if have < (1 if single else args.gpus):
        raise SystemExit(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs, this node exposes {have}; not running "
                         f"(a run on fewer devices would not be an N = {args.gpus} measurement)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs-per-gpu", type=int, default=N_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="wall time of the CPU-baseline sample (default ~10 s)")
    ap.add_argument("--generic", action="store_true",
                    help="time the generic kernels instead of the config-specialised ones (nsg_specialize)")
    ap.add_argument("--no-hbm-resident", action="store_true", help="skip the second roofline figure (2^24 envs)")
    ap.add_argument("--no-all-configs", action="store_true", help="skip config.baseline_configs_at_own_size (C2 / C3 / C4 at their own sizes)")
    ap.add_argument("--dump-shards", default=None, metavar="DIR",
                    help="after the timed steps every rank writes its shard's final rows, rank 0 the gathered returns (sharding-invariance test)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args, sys.argv[1:])          # does not return

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); they must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the product path has no CPU fallback")
    # NSG_BENCH_SINGLE_DEVICE=1 + NSG_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a one-GPU box
    # (every rank on cuda:0, collectives through gloo); the driver's real runs use one GPU per rank + RCCL
    single = os.environ.get("NSG_BENCH_SINGLE_DEVICE") == "1"
    backend = os.environ.get("NSG_BENCH_BACKEND", "nccl")
    if not single and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) on this node")
    dev_index = 0 if single else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device(f"cuda:{dev_index}")
    dist = None
    # NSG_BENCH_FORCE_DIST=1: a single rank still opens its process group, so that the barriers, the all-gather of returns and the
    # max-over-ranks reduction of an N = 1 run go through RCCL - the one way a one-GPU box can exercise that library at all
    if world > 1 or os.environ.get("NSG_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from ns_gym_amd import make
    from ns_gym_amd.distributed import all_gather_returns
    from ns_gym_amd.schedulers import ContinuousScheduler
    from ns_gym_amd.update_functions import IncrementUpdate
    from ns_gym_amd.vec_env import VecNSEnv

    n = args.envs_per_gpu
    def make_env(specialize):
        return VecNSEnv(make("CartPole-v1"), {"masspole": IncrementUpdate(ContinuousScheduler(), k=0.1)}, n,
                        change_notification=True, delta_change_notification=True, track_returns=True, device=dev,
                        specialize=specialize)

    # the product path for a long run: kernels compiled for THIS wrapper configuration (hiprtc, ~1 s, same
    # results bit for bit - tests/test_gpu_specialized.py); --generic times the precompiled generic kernels
    from ns_gym_amd._lib import NsgError

    try:
        env = make_env(not args.generic)
    except NsgError as e:   # no runtime compiler on this box: the precompiled generic kernels are the product path
        if args.generic or "nsg_specialize" not in str(e):
            raise
        print(f"bench.py: {e}; timing the generic kernels", file=sys.stderr)
        args.generic = True
        env = make_env(False)
    kernels_used = env.kernels
    # env i of the whole job is seeded base_seed + global index: results do not depend on the sharding
    env.reset(seed=torch.arange(rank * n, (rank + 1) * n, dtype=torch.int64).numpy().astype("uint64"))
    g = torch.Generator(device=dev)
    g.manual_seed(123 + rank)
    # synthetic random actions, a function of (pool slot, GLOBAL env index): what env i is fed does not depend on the sharding
    from ns_gym_amd.distributed import global_actions

    pool = [global_actions(k, rank * n, (rank + 1) * n, 2, device=dev) for k in range(8)]

    def barrier():
        if dist is not None:
            dist.barrier()

    # Setup (untimed, before the W warm-up steps): one pass over the 8-entry action pool.  The timed loop cycles through the pool;
    # an entry no launch of this process has read yet costs its first launch ~4 us of address-translation misses at the start
    # of every workgroup (rocprofv3 kernel trace of `--steps 20 --warmup 5`: 28.4 / 27.5 / 27.8 us for the launches that read
    # pool[5..7] for the first time, 24.0-24.4 us for the others; tools/trace_gaps.py).  A policy kernel that has just WRITTEN
    # its actions leaves their pages translated, so the steady state the metric is about is the warm one.  Reported as
    # config.setup_steps.
    SETUP_STEPS = len(pool)
    if dist is not None:
        all_gather_returns(env, sizes=[n] * world)  # warm the RCCL communicator
    env.episode_returns()   # warm-up of the end-of-rollout read-out (its first use loads torch's elementwise kernels: ~50 ms once)
    e0 = torch.cuda.Event(enable_timing=True)   # HIP events on the stream the kernels are enqueued on
    e1 = torch.cuda.Event(enable_timing=True)
    ew = torch.cuda.Event()

    def drain(ev):
        """torch.cuda.synchronize(), entered only once the stream has reached `ev`: polling the event costs a few us where a
        blocking wait on an idle-going device costs tens - which a 20-step window (0.5 ms) would carry as 5-10 % of its time."""
        while not ev.query():
            pass
        torch.cuda.synchronize()

    for k in range(SETUP_STEPS):
        env.step(pool[k])
    for k in range(args.warmup):
        env.step(pool[k % 8])
    ew.record()
    drain(ew)
    barrier()
    torch.cuda.synchronize()
    # At N > 1 the barrier is an RCCL collective that ends in a BLOCKING stream wait: the device sits idle for tens of us before
    # this process is back, and the next launches would pay for that gap (see drain()).  Two untimed steps and a polled drain
    # put the device back where the W warm-up steps left it; the ranks leave this point within a few us of each other.
    REWARM_STEPS = 2
    for k in range(REWARM_STEPS):
        env.step(pool[(args.warmup + k) % 8])
    ew.record()
    drain(ew)
    # The timed region is the K steps and nothing else: the opening event is enqueued (on the idle stream) before the clock is
    # read, and the clock is read again the moment the event behind the K-th step has completed - the device has drained - with the
    # closing torch.cuda.synchronize() + barrier right behind it.  (A synchronize() on a device that is already idle takes ~20 us
    # of host time and the event record ~6: in the driver's 20-step window - 0.46 ms of kernels - they were 6 % of the quotient,
    # tools/window_probe.py.  The clock read AFTER the synchronize is kept as config.ms_per_step_clock_after_synchronize.)
    e0.record()
    t0 = time.perf_counter()
    for k in range(args.steps):
        env.step(pool[k % 8])
    e1.record()
    dt_enqueue = time.perf_counter() - t0   # the host loop alone: how fast THIS rank's Python issues launches (a rank starved of host cores shows here)
    while not e1.query():
        pass
    dt = time.perf_counter() - t0      # this rank's K steps, device drained; the MAX over ranks below is the job's time
    torch.cuda.synchronize()
    dt_after_sync = time.perf_counter() - t0
    barrier()                          # closing bracket (its own latency - an RCCL collective - is not part of the steps)
    torch.cuda.synchronize()
    kern_ms = e0.elapsed_time(e1) / args.steps   # average launch duration incl. back-to-back gap

    # The job's ONE exchange: every rank's episode returns gathered at the END of a rollout (per rollout, not per step - a
    # rollout is thousands of steps; inside the driver's 20-step window it would be a third of the time at N = 8).  Timed on
    # its own and reported as config.returns_gather_ms, with barriers on both sides like the steps.
    # Two schedules of the same exchange, both timed: RCCL's all-gather (its own choice of algorithm) and the direct one - every
    # shard sent to each peer over that peer's own xGMI link (ns_gym_amd.distributed.all_gather_returns_direct).  NSG_BENCH_ALLGATHER
    # = "rccl" (default) | "direct" picks the one whose time is charged as returns_gather_ms / value_incl_gather.
    from ns_gym_amd.distributed import all_gather_returns_direct, verify_gather

    def timed_gather(fn):
        barrier()
        torch.cuda.synchronize()
        tg = time.perf_counter()
        out = fn(env, sizes=[n] * world) if dist is not None else env.episode_returns()[0]
        torch.cuda.synchronize()
        barrier()
        return out, (time.perf_counter() - tg) * 1e3

    which = os.environ.get("NSG_BENCH_ALLGATHER", "rccl")
    if which not in ("rccl", "direct"):
        raise SystemExit("bench.py: NSG_BENCH_ALLGATHER must be 'rccl' or 'direct'")
    gathered, gather_ms_rccl = timed_gather(all_gather_returns)
    gather_ms_direct, direct_equal = None, None
    if dist is not None:
        all_gather_returns_direct(env, sizes=[n] * world)          # first use: connection set-up of the point-to-point channels
        g2, gather_ms_direct = timed_gather(all_gather_returns_direct)
        direct_equal = bool(torch.equal(g2, gathered))
    gather_ms = gather_ms_direct if (which == "direct" and gather_ms_direct is not None) else gather_ms_rccl
    # the gathered CONTENT, not just its length: this rank's slice is its local tensor, and the checksums agree across ranks
    gather_verified = verify_gather(gathered, env.episode_returns()[0], rank * n) and (direct_equal is not False)

    if args.dump_shards:   # every row a sharded job must reproduce whatever its world size
        import numpy as np

        os.makedirs(args.dump_shards, exist_ok=True)
        torch.cuda.synchronize()
        np.savez(os.path.join(args.dump_shards, f"world{world}_rank{rank}.npz"), lo=rank * n, hi=(rank + 1) * n,
                 obs=env.state.cpu().numpy(), phys=env.phys.cpu().numpy(), t=env.t.cpu().numpy(), theta=env.theta.cpu().numpy(),
                 episode=env.buf["episode"].cpu().numpy(), last_length=env.buf["last_length"].cpu().numpy(),
                 reward=env.reward.cpu().numpy(), terminated=env.terminated.cpu().numpy(), truncated=env.truncated.cpu().numpy(),
                 delta_change=env.gt_delta_change.cpu().numpy())
        if rank == 0:
            np.save(os.path.join(args.dump_shards, f"world{world}_gathered_returns.npy"), gathered.cpu().numpy())

    episodes_rank0 = env.counters()["episodes"]
    # secondary figure (not `value`): the same env-steps through nsg_rollout, K = 64 fused steps per launch
    # with the persistent rows held in registers (callers that supply K actions at once: planners' rollouts)
    K = 64
    acts64 = torch.stack([pool[k % 8] for k in range(K)])
    REC = ("obs", "reward", "terminated", "truncated", "env_change", "delta_change")
    env.rollout(acts64, record=REC)   # same outputs as the timed calls: their [K, N] slices are allocated here, not in the timed region
    torch.cuda.synchronize()
    r0 = torch.cuda.Event(enable_timing=True)
    r1 = torch.cuda.Event(enable_timing=True)
    r0.record()
    for _ in range(4):
        env.rollout(acts64, record=REC)
    r1.record()
    torch.cuda.synchronize()
    rollout_rate = 4 * K * float(n) / (r0.elapsed_time(r1) * 1e-3)

    # for the record (not `value`): the other kernel flavour on the same workload, 300 launches
    other_us = None
    try:
        other = make_env(args.generic)
        other.reset(seed=torch.arange(rank * n, (rank + 1) * n, dtype=torch.int64).numpy().astype("uint64"))
        for k in range(30):
            other.step(pool[k % 8])
        other_us = other.time_steps(pool[0], 300) * 1e3
        other.close()
    except Exception:   # for the record only
        pass

    own_size = None
    if world == 1 and not args.no_all_configs and not args.generic:
        try:
            own_size = baseline_configs_at_own_size(dev, kern_ms * 1e3, rollout_rate, n)
        except Exception as e:   # a side table must never cost the run its headline line
            own_size = {"error": f"{type(e).__name__}: {e}"[:200]}

    # second roofline figure (not `value`): the same kernel on a batch whose rows cannot live in the Infinity Cache
    # (2^24 envs = 2.5 GB of rows against 256 MiB), i.e. streamed from HBM on every step.  Rank 0 of an N = 1 run only.
    hbm = None
    if world == 1 and not args.no_hbm_resident and not args.generic and n != N_HBM_RESIDENT:
        try:
            env.close()
            big = VecNSEnv(make("CartPole-v1"), {"masspole": IncrementUpdate(ContinuousScheduler(), k=0.1)}, N_HBM_RESIDENT,
                           change_notification=True, delta_change_notification=True, track_returns=True, device=dev, specialize=True)
            big.reset(seed=0)
            big_kernels = big.kernels
            ab = torch.randint(0, 2, (N_HBM_RESIDENT,), dtype=torch.int32, device=dev, generator=g)
            for _ in range(20):
                big.step(ab)
            # hipEvents on the launch stream around back-to-back launches; three repetitions of 100: the figure is their median
            # and the spread stays visible (boxes differ by up to 15 % on this one, a box's own repetitions by ~1 %)
            reps_us = sorted(big.time_steps(ab, 100) * 1e3 for _ in range(3))
            big_us = reps_us[1]
            big.close()
            ach = BYTES_PER_ENV_STEP * N_HBM_RESIDENT / (big_us * 1e-6) / 1e9
            hbm = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                   "envs": N_HBM_RESIDENT, "avg_launch_us": big_us, "launches": 100, "repetitions": 3, "repetitions_us": reps_us,
                   "algorithmic_bytes_per_env_step": BYTES_PER_ENV_STEP, "traffic": None,
                   "kernels": big_kernels,
                   "note": "same kernel and config as `roofline`, 2^24 envs: every row streams from HBM each step"}
        except Exception as e:     # e.g. not enough device memory next to another tenant; never fatal for the headline line
            hbm = {"error": f"{type(e).__name__}: {e}"[:200]}

    t = torch.tensor([dt, gather_ms, kern_ms, gather_ms_rccl, gather_ms_direct or 0.0, dt_enqueue], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max, gather_ms_max, kern_ms_max, gather_ms_rccl_max, gather_ms_direct_max, dt_enqueue_max = (float(x) for x in t.tolist())
    total_env_steps = float(n) * world * args.steps
    value = total_env_steps / dt_max
    # the same job with its one exchange charged to it: this window's K steps + the gather, and a T = 1000 rollout + the gather
    value_incl_gather = total_env_steps / (dt_max + gather_ms_max * 1e-3)
    value_incl_gather_t1000 = float(n) * world * 1000 / (dt_max / args.steps * 1000 + gather_ms_max * 1e-3)

    if rank == 0:
        achieved, peak, multi = job_roofline(value, world, n, kern_ms, kern_ms_max)
        out = {
            "metric": "non-stationary env-steps/sec",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "CartPole-v1 + masspole IncrementUpdate(+0.1)/ContinuousScheduler (BASELINE C1/C5 config), "
                            f"{n} envs per GPU, random actions, next-step autoreset, episode returns tracked",
                "envs_per_gpu": n, "total_envs": n * world, "setup_steps": SETUP_STEPS, "rewarm_steps": REWARM_STEPS,
                "parallelism": f"env-sharded x{world}, no per-step collective; 1 all-gather of episode returns at rollout end (returns_gather_ms, outside the K timed steps)"
                               if world > 1 else "single GPU",
                "episodes_finished_rank0": episodes_rank0,
                "gathered_returns": int(gathered.numel()), "returns_gather_ms": gather_ms_max,
                "gather_verified": bool(gather_verified), "returns_gather_schedule": which if dist is not None else "local read-out",
                "returns_gather_ms_rccl_allgather": gather_ms_rccl_max,
                "returns_gather_ms_direct_p2p": gather_ms_direct_max if dist is not None else None,
                # the host side of the K timed steps, per rank: Python launch loop only (before the drain)
                "rank0_host_loop_steps_per_s": args.steps / dt_enqueue, "slowest_rank_host_loop_steps_per_s": args.steps / dt_enqueue_max,
                "rank0_avg_launch_us": kern_ms * 1e3, "slowest_rank_avg_launch_us": kern_ms_max * 1e3,
                "collectives": (f"torch.distributed backend {dist.get_backend()}, {world} rank(s)" if dist is not None else "none (single process)"),
                "value_incl_gather": value_incl_gather, "value_incl_gather_T1000": value_incl_gather_t1000,
                "ms_per_step_clock_after_synchronize": dt_after_sync / args.steps * 1e3,
                "timed_region": "clock read after the opening event is enqueued and again when the event behind the K-th step has completed "
                                "(device drained); the closing synchronize() + barrier follow immediately",
                "actions": "counter-based uniform draws per (pool slot, global env index): independent of the sharding",
                "rollout_k64_env_steps_per_sec_per_gpu": rollout_rate,
                # which code object ran: "config-specialised (prebuilt)" = the unit shipped with the library, built and inspected at
                # library build time (ns_gym_amd/prebuilt/resource_usage.txt); "(hiprtc)" = compiled on this box by the runtime compiler
                "kernels": kernels_used,
                ("specialised_kernel_avg_launch_us" if args.generic else "generic_kernel_avg_launch_us"): other_us,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": peak, "unit": "GB/s",
                "frac": achieved / peak, "traffic": _pmc_traffic(n),
                # `frac` prices the kernel: bytes per launch over the HIP-event average launch duration (`avg_launch_us`).  The same
                # bytes over the line's own `ms_per_step` (host clock around the K steps, max over ranks):
                "frac_from_ms_per_step": BYTES_PER_ENV_STEP * value / 1e9 / (HBM_PEAK_GBS * world),
                "traffic_source": "profiles/pmc_traffic.json (static: rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE passes of an "
                                  "earlier run of this kernel, scaled per env; not measured in this run)",
                "residency": f"{n} envs = {n * 150 / 1e6:.0f} MB of rows: "
                             + ("inside the 256-MiB Infinity Cache - `achieved` is a rate out of that cache, not out of HBM; "
                                "see roofline_hbm_resident" if n * 150 < 256 << 20 else "beyond the 256-MiB Infinity Cache"),
                "kernel": "nsg::step_kernel<CARTPOLE,false>" if args.generic else "nsg_spec_step (nsg::step_body<CARTPOLE,false>, config folded)",
                "avg_launch_us": kern_ms * 1e3,
                "algorithmic_bytes_per_env_step": BYTES_PER_ENV_STEP,
            },
        }
        if multi is not None:
            out["roofline"].update(multi)
            out["roofline"]["basis"] = (f"aggregate: {BYTES_PER_ENV_STEP} B x value (all {world} ranks, max-over-ranks time) against "
                                        f"{world} x {HBM_PEAK_GBS:.0f} GB/s; avg_launch_us is rank 0's kernel")
        if own_size is not None:
            out["config"]["baseline_configs_at_own_size"] = own_size
        if hbm is not None:
            out["roofline_hbm_resident"] = hbm
        if not args.no_cpu_baseline and world == 1:   # the CPU baseline is reported on rank 0 at N = 1 only
            # the GPU box exposes 256 logical CPUs but one GPU's share is 16 cores
            threads = min(16, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1))
            cpu_baseline(1 << 14, 8, threads)  # warm the OpenMP pool / page in the oracle
            # bounded sample of the same workload: 2^18 envs, step count sized from a short probe so the
            # timed sample is ~10 s of wall time on the host share (never more than 20 s)
            n_cpu = 1 << 18
            rate, _ = cpu_baseline(n_cpu, 40, threads)
            steps_cpu = int(min(max(args.cpu_seconds * rate / n_cpu, 40), 2 * args.cpu_seconds * rate / n_cpu, 200000))
            v, secs = cpu_baseline(n_cpu, steps_cpu, threads)
            out["cpu_baseline"] = {
                "value": v, "unit": "env-steps/s", "cores": threads, "kind": "port",
                "sample": f"oracle C port (OpenMP, {threads} threads), {n_cpu} envs x {steps_cpu} steps of the same config, "
                          f"{secs:.1f} s wall",
                # for context: the shape the reference's path has today - one Python wrapper object per env, one step()
                # call per env per step, dict observations (oracle/python_loop.py), 1 core
                "python_object_loop": _safe(_python_object_loop),
                # the same loop in one worker PROCESS per host core of this GPU's share, rates summed
                "python_object_loop_all_cores": _safe(_python_object_loop_all_cores, threads),
                # what a NumPy user would write: the same config as arrays over envs, one core
                "numpy_vectorised": _safe(_numpy_vectorised),
            }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _safe(fn, *a):
    """Context figures must not cost the run its line."""
    try:
        return fn(*a)
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"[:200]}


def _python_object_loop():
    import numpy as np

    from oracle import python_loop as PL

    envs = [PL.make_c1() for _ in range(64)]
    for i, e in enumerate(envs):
        e.reset(seed=i)
    acts = np.random.default_rng(123).integers(2, size=(8, 64))
    PL.run_loop(envs, acts, 20)
    t0 = time.perf_counter()
    n = PL.run_loop(envs, acts, 600)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "env-steps/s", "cores": 1,
            "sample": f"64 wrapper objects x 600 steps of the same config in {dt:.1f} s"}


def _numpy_vectorised():
    from oracle import numpy_vec as NV

    n = 1 << 16
    rate, _ = NV.time_c1(n, 10)
    steps = int(min(max(3.0 * rate / n, 10), 2000))     # ~3 s of wall time
    v, secs = NV.time_c1(n, steps)
    return {"value": v, "unit": "env-steps/s", "cores": 1,
            "sample": f"oracle/numpy_vec.py, {n} envs x {steps} steps of the same config in {secs:.1f} s"}


def _python_object_loop_all_cores(workers):
    """`workers` child processes (plain children of this one, started after the GPU work is done), each running the object
    loop on its own 64 wrapper objects; the sum of their rates."""
    import subprocess

    env = dict(os.environ, OMP_NUM_THREADS="1", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.python_loop", "600"], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.DEVNULL, text=True) for _ in range(workers)]
    rates = []
    for p in procs:
        out, _ = p.communicate(timeout=300)
        if p.returncode == 0:
            rates.append(float(out.strip().splitlines()[-1]))
    return {"value": sum(rates), "unit": "env-steps/s", "cores": len(rates),
            "sample": f"{len(rates)} processes x 64 wrapper objects x 600 steps, {time.perf_counter() - t0:.1f} s wall incl. interpreter start-up"}


def _pmc_traffic(n_envs):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), or None.  The passes were
    taken at the default 2^20 envs per GPU; the figure is per env-step, scaled to this run's batch."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            doc = json.load(f)
            key = next(k for k in ("r03_step_kernel_cartpole_specialised_2p20", "r02_step_kernel_cartpole_specialised_2p20",
                                   "step_kernel_cartpole_specialised") if k in doc)
            return doc[key]["bytes_per_env_step"] * n_envs
    except Exception:
        return None


if __name__ == "__main__":
    main()
