"""Device evaluation of the θ-engine and of NumPy-compatible streams through the C-ABI
(nsg_theta_trace / nsg_rng_fill).  Used by `Scheduler.__call__`, `UpdateFn.__call__` and the
known-answer parity tests."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _abi as A
from . import _lib
from .base import Scheduler, UpdateDistributionFn, UpdateFn
from .envs import make
from .spec import compile_config


def _dev():
    if not torch.cuda.is_available():
        raise _lib.NsgError("the θ-engine runs on the GPU only (no CPU path)")
    return torch.device(f"cuda:{torch.cuda.current_device()}")


def rng_fill(kind: int, seeds, count: int, spawn_key: int = -1):
    """kind 0 raw uint64, 1 Generator.random(), 2 Generator.standard_normal(); returns
    (out[count, n], state[4, n]) as NumPy arrays."""
    lib, dev = _lib.load(), _dev()
    s = np.asarray(seeds, dtype=np.uint64).reshape(-1)
    n = s.size
    sd = torch.from_numpy(s.view(np.int64)).to(dev)
    out = torch.zeros((max(count, 1), n), dtype=torch.int64 if kind == 0 else torch.float64, device=dev)
    st = torch.zeros((n, 4), dtype=torch.int64, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(lib.nsg_rng_fill(kind, sd.data_ptr(), n, spawn_key, count, out.data_ptr(), st.data_ptr(), stream), "nsg_rng_fill")
    torch.cuda.synchronize(dev)
    o = out[:count].cpu().numpy()
    return (o.view(np.uint64) if kind == 0 else o), st.cpu().numpy().view(np.uint64).T.copy()  # [4, n]


def theta_trace(fn: UpdateFn, theta0, t0: int = 0, T: int = 1, n: int = 1, seeds=None, state: dict | None = None):
    """Drive one (scheduler, update fn) pair for t = t0..t0+T-1 with θ fed back, on `n` lanes.
    Returns (theta[T, n] or [T, 3, n], fired[T, n], delta[T, n]) as NumPy arrays.

    `state`: a dict owned by the caller (empty on the first call).  When given, the object's whole state -
    update-fn stream, list cursor / prev_time, stochastic scheduler stream and transition_time - is taken
    from it and written back to it (device tensors), so successive calls continue like successive calls of
    the reference's stateful objects."""
    lib, dev = _lib.load(), _dev()
    dist = isinstance(fn, UpdateDistributionFn)
    nd = len(theta0) if dist else 0          # 3: FrozenLake / Bridge support, 4: CliffWalking
    env = (make("CliffWalking-v1") if nd == 4 else make("FrozenLake-v1")) if dist else make("CartPole-v1")
    kw = {"initial_prob_dist": [1.0] + [0.0] * (nd - 1)} if dist else {}
    # a sampled (Custom) scheduler must hold an answer for every t this trace asks about
    cfg, tables, _, _ = compile_config(env, {"P" if dist else "gravity": fn}, table_horizon=int(t0) + int(T) + 1, **kw)
    h = C.c_void_p()
    _lib.check(lib.nsg_create(C.byref(cfg), tables, len(tables), max(n, 1), C.byref(h)), "nsg_create")
    try:
        th0 = np.array(np.broadcast_to(np.asarray(theta0, dtype=np.float64), (n, nd) if dist else (n,)))  # writable copy
        d_th0 = torch.from_numpy(th0).to(dev)
        rng = state.get("rng") if state else None
        if cfg.params[0].uses_rng and rng is None:
            sd = np.asarray(seeds if seeds is not None else [cfg.params[0].fn_seed] * n, dtype=np.uint64)
            _, st = rng_fill(0, sd, 0)
            rng = torch.from_numpy(np.ascontiguousarray(st.T).view(np.int64)).to(dev)  # [n, 4] records
        th = torch.zeros((T, nd, n) if dist else (T, n), dtype=torch.float64, device=dev)
        fired = torch.zeros((T, n), dtype=torch.uint8, device=dev)
        delta = torch.zeros((T, n), dtype=torch.float64, device=dev)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if state is None:
            _lib.check(lib.nsg_theta_trace(h, 0, n, int(t0), int(T), d_th0.data_ptr(), rng.data_ptr() if rng is not None else None,
                                           th.data_ptr(), fired.data_ptr(), delta.data_ptr(), stream), "nsg_theta_trace")
        else:
            resume = bool(state.get("started"))
            for key, shape in (("cursor", (n,)), ("sched_next", (n,))):
                if key not in state:
                    state[key] = torch.zeros(shape, dtype=torch.int32, device=dev)
            if "sched_rng" not in state:
                state["sched_rng"] = torch.zeros((n, 4), dtype=torch.int64, device=dev)
            if rng is not None:
                state["rng"] = rng
            ts = A.TraceState(rng=rng.data_ptr() if rng is not None else None, cursor=state["cursor"].data_ptr(),
                              sched_rng=state["sched_rng"].data_ptr(), sched_next=state["sched_next"].data_ptr(),
                              resume=1 if resume else 0)
            _lib.check(lib.nsg_theta_trace_stateful(h, 0, n, int(t0), int(T), d_th0.data_ptr(), C.byref(ts), th.data_ptr(),
                                                    fired.data_ptr(), delta.data_ptr(), stream), "nsg_theta_trace_stateful")
            state["started"] = True
        torch.cuda.synchronize(dev)
        f = fired.cpu().numpy()
        if (f == 0xFF).any():   # the reference raises from LCBoundedDistrubutionUpdate._update (distribution.py:178-182)
            raise ValueError(f"Could not find a Lipschitz-continuous update after {int(1e5)} attempts (L={getattr(fn, 'L', None)})")
        if (f == 0xFE).any():
            raise ValueError("CustomScheduler asked about a t beyond the horizon its event function was sampled over")
        return th.cpu().numpy(), f, delta.cpu().numpy()
    finally:
        lib.nsg_destroy(h)


def schedule_fires(scheduler: Scheduler, ts, state: dict | None = None) -> np.ndarray:
    """Scheduler.__call__ for each t in `ts` (device).  Deterministic schedulers are pure functions of t
    (one trace over [min, max]); a stochastic scheduler is called once per t, in order, on its `state`."""
    from .update_functions import NoUpdate

    ts = [int(t) for t in ts]
    if not ts:
        return np.zeros(0, dtype=bool)
    if state is not None:
        return np.array([bool(theta_trace(NoUpdate(scheduler), 0.0, t0=t, T=1, state=state)[1][0, 0]) for t in ts])
    lo, hi = min(ts), max(ts)
    _, fired, _ = theta_trace(NoUpdate(scheduler), 0.0, t0=lo, T=hi - lo + 1)
    return np.array([bool(fired[t - lo, 0]) for t in ts])
