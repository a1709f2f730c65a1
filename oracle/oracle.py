"""TEST INFRASTRUCTURE ONLY — ctypes binding of oracle/libnsgym_oracle.so (the scalar C
restatement of the reference's hot path, see nsgym_oracle.c).

`OracleVecEnv` keeps N env instances in NumPy struct-of-arrays with exactly the data format
of include/nsgym_hip.h so that tests compare the HIP library's device buffers with these
arrays field by field.  It uses `ns_gym_amd.spec.compile_config` only as the config *format*
encoder; no arithmetic of the product is involved.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from ns_gym_amd import _abi as A
from ns_gym_amd.spec import compile_config

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libnsgym_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "nsgym_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_sizeof_config.restype = C.c_size_t
        _lib.orc_sizeof_buffers.restype = C.c_size_t
        assert _lib.orc_sizeof_config() == C.sizeof(A.Config)
        assert _lib.orc_sizeof_buffers() == C.sizeof(A.Buffers)
    return _lib


_NP = {C.c_double: np.float64, C.c_int32: np.int32, C.c_uint8: np.uint8, C.c_uint64: np.uint64,
       C.c_float: np.float32, C.c_uint32: np.uint32}

PHYS_DIM = {A.ENV_CARTPOLE: 4, A.ENV_PENDULUM: 2, A.ENV_ACROBOT: 4, A.ENV_MOUNTAINCAR: 2,
            A.ENV_MOUNTAINCAR_CONT: 2, A.ENV_FROZENLAKE: 0, A.ENV_CLIFFWALKING: 0, A.ENV_BRIDGE: 0}
OBS_DIM = {A.ENV_CARTPOLE: 4, A.ENV_PENDULUM: 3, A.ENV_ACROBOT: 6, A.ENV_MOUNTAINCAR: 2,
           A.ENV_MOUNTAINCAR_CONT: 2, A.ENV_FROZENLAKE: 1, A.ENV_CLIFFWALKING: 1, A.ENV_BRIDGE: 1}


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleVecEnv:
    def __init__(self, env, tunable_params, num_envs, **kwargs):
        self._tp, self._kwargs = tunable_params, dict(kwargs)
        self.cfg, self.tables, self.spec, self.param_names = compile_config(env, tunable_params, **kwargs)
        self.tab = np.frombuffer(self.tables, dtype=np.uint8).copy()
        self.N = N = int(num_envs)
        et = self.cfg.env_type
        P = self.cfg.n_params
        self.is_fl = et in A.GRID_ENVS
        self.nd = A.N_DIST.get(et, 3)
        rows = self.nd * P if self.is_fl else P
        z = lambda shape, dt: np.zeros(shape, dtype=dt)  # noqa: E731
        self.a = {
            "phys": z((max(PHYS_DIM[et], 1), N), np.float64), "cell": z(N, np.int32),
            "theta": z((max(rows, 1), N), np.float64), "table_prob": z((4, N), np.float64), "derived": z((4, N), np.float64), "t": z(N, np.int32), "t_fork": z(N, np.int32), "status": z(N, np.uint8), "episode": z(N, np.int32),
            "rng_env": z((N, 4), np.uint64), "rng_upd": z((max(P, 1), N, 4), np.uint64), "rng_sched": z((max(P, 1), N, 4), np.uint64),
            "sched_next": z((max(P, 1), N), np.int32),
            "cursor": z((max(P, 1), N), np.int32), "obs": z((N, OBS_DIM[et]), np.float32),
            "reward": z(N, np.float32), "terminated": z(N, np.uint8), "truncated": z(N, np.uint8),
            "env_change": z((max(P, 1), N), np.uint8), "delta_change": z((max(P, 1), N), np.float32),
            "violation": z((max(P, 1), N), np.uint8), "prob": z(N, np.float32), "ep_return": z(N, np.float32), "ep_length": z(N, np.int32),
            "last_return": z(N, np.float32), "last_length": z(N, np.int32),
            "counters": z((A.CNT_COUNT, A.CNT_SHARDS), np.uint64), "done_bits": z((N + 63) // 64, np.uint64),
        }
        self.bufs = A.Buffers(**{k: _ptr(v) for k, v in self.a.items()})
        lib().orc_init_streams(C.byref(self.cfg), C.byref(self.bufs), C.c_int64(N), None)
        self.action_is_float = et in (A.ENV_PENDULUM, A.ENV_MOUNTAINCAR_CONT)

    def reset(self, seed=None, mask=None):
        seeds = None
        if seed is not None:
            seeds = (np.arange(self.N, dtype=np.uint64) + np.uint64(seed)) if np.isscalar(seed) \
                else np.asarray(seed, dtype=np.uint64)
            assert seeds.shape == (self.N,)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        lib().orc_reset(C.byref(self.cfg), _ptr(self.tab), C.byref(self.bufs), C.c_int64(self.N), _ptr(seeds), _ptr(m))
        return self.a

    def step(self, actions):
        dt = np.float32 if self.action_is_float else np.int32
        act = np.ascontiguousarray(actions, dtype=dt)
        assert act.shape == (self.N,)
        lib().orc_step(C.byref(self.cfg), _ptr(self.tab), C.byref(self.bufs), C.c_int64(self.N), _ptr(act))
        return self.a

    def fork(self, theta_mode=0, entropy=0):
        """get_planning_env()/__deepcopy__ twin of VecNSEnv.fork."""
        kw = dict(self._kwargs)
        kw["is_sim_env"] = True
        dst = OracleVecEnv(self.spec, self._tp, self.N, **kw)
        lib().orc_fork(C.byref(self.cfg), C.byref(self.bufs), C.byref(dst.cfg), C.byref(dst.bufs), C.c_int64(self.N),
                       C.c_uint64(entropy), int(theta_mode))
        return dst

    def seed_streams(self, seeds, which=0):
        s = np.ascontiguousarray(seeds, dtype=np.uint64)
        lib().orc_seed_streams(C.byref(self.cfg), C.byref(self.bufs), C.c_int64(self.N), _ptr(s), int(which))

    def step_mt(self, actions, nthreads):
        dt = np.float32 if self.action_is_float else np.int32
        act = np.ascontiguousarray(actions, dtype=dt)
        lib().orc_step_mt(C.byref(self.cfg), _ptr(self.tab), C.byref(self.bufs), C.c_int64(self.N), _ptr(act), int(nthreads))
        return self.a

    def rollout_policy(self, kind, k_steps, data=None, seed=0, index0=0, step0=0, accounts=None):
        """The reference's closed loops restated (orc_rollout_policy): K steps of `action = policy(last observation); step; add`.
        `kind`: A.NSG_POL_*; `data`: the action table / state table / weight rows (NumPy); `accounts`: dict of NumPy arrays
        ret (float64), length (int32), alive (uint8), discount (float64 or None), updated in place.
        Returns (actions[K, N], reward64[K, N], took[K, N])."""
        K, N = int(k_steps), self.N
        acts = np.zeros((K, N), dtype=np.float32 if self.action_is_float else np.int32)
        r64 = np.zeros((K, N), dtype=np.float64)
        took = np.zeros((K, N), dtype=np.uint8)
        d = None
        if data is not None:
            d = np.ascontiguousarray(data)
        n_data = 0 if d is None else (int(d.shape[0]) if kind == A.NSG_POL_LINEAR else int(d.size))
        pol = A.Policy(kind=int(kind), step0=int(step0), seed=int(seed) & ((1 << 64) - 1), index0=int(index0), data=_ptr(d), n_data=n_data,
                       reserved0=0, actions_out=_ptr(acts))
        acc = None
        if accounts is not None:
            disc = accounts.get("discount")
            acc = A.EpisodeAcc(ret=_ptr(accounts["ret"]), length=_ptr(accounts["length"]), alive=_ptr(accounts["alive"]),
                               discount=_ptr(disc), n_discount=0 if disc is None else int(disc.size), reserved0=0)
        rc = lib().orc_rollout_policy(C.byref(self.cfg), _ptr(self.tab), C.byref(self.bufs), C.c_int64(N), C.byref(pol), K,
                                      C.byref(acc) if acc is not None else None, _ptr(r64), _ptr(took))
        assert rc == 0
        return acts, r64, took

    # convenience views -----------------------------------------------------------------
    def state(self):
        return self.a["cell"].copy() if self.is_fl else self.a["obs"].copy()

    def theta(self):
        return self.a["theta"].copy()


def theta_trace(fn, theta0, t0=0, T=1, n=1, seeds=None):
    """Oracle twin of ns_gym_amd.functional.theta_trace: drive one (scheduler, update fn)
    pair for t = t0..t0+T-1 with θ fed back; returns (theta[T,...], fired[T,n], delta[T,n])."""
    from ns_gym_amd.base import UpdateDistributionFn
    from ns_gym_amd.envs import make

    dist = isinstance(fn, UpdateDistributionFn)
    nd = len(theta0) if dist else 0
    env = (make("CliffWalking-v1") if nd == 4 else make("FrozenLake-v1")) if dist else make("CartPole-v1")
    kw = {"initial_prob_dist": [1.0] + [0.0] * (nd - 1)} if dist else {}
    cfg, tables, _, _ = compile_config(env, {"P" if dist else "gravity": fn}, **kw)
    tab = np.frombuffer(tables, dtype=np.uint8).copy()
    th0 = np.ascontiguousarray(np.broadcast_to(np.asarray(theta0, dtype=np.float64), (n, nd) if dist else (n,)))
    rng = None
    if cfg.params[0].uses_rng:
        sd = np.asarray(seeds if seeds is not None else [cfg.params[0].fn_seed] * n, dtype=np.uint64)
        rng = np.zeros((n, 4), dtype=np.uint64)
        scratch = np.zeros((1, n), dtype=np.uint64)
        lib().orc_rng_fill(0, _ptr(sd), n, -1, 0, _ptr(scratch), _ptr(rng))
    th = np.zeros((T, nd, n) if dist else (T, n), dtype=np.float64)
    fired = np.zeros((T, n), dtype=np.uint8)
    delta = np.zeros((T, n), dtype=np.float64)
    lib().orc_theta_trace(C.byref(cfg), _ptr(tab), 0, n, int(t0), int(T), _ptr(th0), _ptr(rng), _ptr(th),
                          _ptr(fired), _ptr(delta))
    return th, fired, delta


def rng_fill(kind, seeds, count, spawn_key=-1):
    seeds = np.asarray(seeds, dtype=np.uint64)
    n = seeds.size
    out = np.zeros((count, n), dtype=np.uint64 if kind == 0 else np.float64)
    st = np.zeros((n, 4), dtype=np.uint64)
    lib().orc_rng_fill(int(kind), _ptr(seeds), n, int(spawn_key), int(count), _ptr(out), _ptr(st))
    return out, st.T.copy()   # [4, n]: state_hi, state_lo, inc_hi, inc_lo
