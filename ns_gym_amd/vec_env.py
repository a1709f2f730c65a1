"""VecNSEnv — N non-stationary env instances stepped by one fused HIP launch.

Host-side mirror of the reference's wrapper interface for the hot path
(NSWrapper / NSClassicControlWrapper / NSFrozenLakeWrapper: ns_gym/base.py:206-502,
ns_gym/wrappers/classic_control.py:15-109, ns_gym/wrappers/toy_text.py:265-399), batched:
env i behaves like one reference wrapper instance reset with `seed + i`.

PyTorch owns every device buffer (the tensors below); the library (C-ABI, include/nsgym_hip.h)
only enqueues kernels on the current stream.  No host synchronisation in `step`.
"""
from __future__ import annotations

import ctypes as C
import warnings

import numpy as np
import torch

from . import _abi as A
from . import _lib
from .spec import compile_config

_TORCH_DT = {C.c_double: torch.float64, C.c_int32: torch.int32, C.c_uint8: torch.uint8,
             C.c_uint64: torch.int64,  # bit pattern; torch has no first-class uint64 arithmetic
             C.c_float: torch.float32}


_NP_DT = {C.c_double: "<f8", C.c_int32: "<i4", C.c_uint8: "u1", C.c_uint64: "<u8", C.c_float: "<f4"}


class ConstraintViolationWarning(Warning):
    """Issued when updates were rejected by the physical-constraint checker
    (ns_gym/wrappers/classic_control.py:9-12)."""


class SpecializationUnavailableWarning(RuntimeWarning):
    """A batch large enough to be specialised by default fell back to the generic kernels (no runtime compiler)."""


_AUTO_SPECIALIZE_MIN_ENVS = 1 << 16   # below this a launch is latency-bound and the compile is not worth a second


class VecNSEnv:
    """Vectorised non-stationary environment (duck-types gymnasium.vector.VectorEnv).

    Args mirror NSWrapper.__init__ (ns_gym/base.py:222-232) plus `num_envs`; FrozenLake adds
    `initial_prob_dist`, `modified_rewards` (ns_gym/wrappers/toy_text.py:282-294).
    """

    def __init__(self, env, tunable_params: dict, num_envs: int, change_notification: bool = False,
                 delta_change_notification: bool = False, in_sim_change: bool = False, scalar_reward: bool = True,
                 persistent_params: bool = False, track_returns: bool = False, device=None, is_sim_env: bool = False,
                 violation_mask: bool = False, specialize: bool | None = None, autoreset: bool = True, libm_exact: bool = False,
                 _compiled=None, **kwargs):
        """`specialize=True` compiles config-specialised step / rollout kernels for this batch (hiprtc, once per
        distinct configuration, ~0.6 s; the code objects persist in `NSG_SPEC_CACHE=<dir>`, default the user's cache directory): same results bit
        for bit, 10-35 % less time per step.  `False`: the precompiled generic kernels.  `None` (default): specialise
        batches of >= 65 536 envs when the runtime compiler is available, silently stay generic otherwise.

        `libm_exact=True` (classic control): the integrators and the θ-engine evaluate sin / cos / exp and every `x ** 2` the reference
        hands to libm's pow with libm's own algorithms and roundings (glibc 2.35's FMA builds - what `np.sin` / `np.cos` / a scalar
        `** 2` resolve to in the reference), so float64 state, observation, reward and θ EQUAL the reference's, bit for bit, for as
        long as the batch is stepped - also where an unstable or chaotic plant (a balanced CartPole, Acrobot) would otherwise amplify
        the last ulp of the kernels' own < 1-ulp sincos into a different trajectory a few hundred steps later.  It runs on the
        batch's specialised unit (implies `specialize=True`; raises if no unit can be had) and costs +4 % (CartPole) to x 1.9 (Acrobot)
        per step.  Planning copies inherit it.

        `autoreset=True` (default): gymnasium's next-step vector autoreset - the step after an episode ended resets that env
        (reward 0, flags clear, relative_time 0, streams continue).  `autoreset=False`: nothing resets inside `step()`; a finished
        env keeps stepping exactly like the reference's single wrappers, which forward to gymnasium whatever `done` said
        (ns_gym/base.py:313): CartPole integrates on and pays 0.0 from its second terminated step, TimeLimit keeps reporting
        `truncated`, FrozenLake's terminal cell self-loops, θ and `relative_time` keep evolving - until `reset()` is called."""
        self.lib = _lib.load()
        self._row_cache = {}
        if not torch.cuda.is_available():
            raise _lib.NsgError("VecNSEnv needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.index is None:
            self.device = torch.device(f"cuda:{torch.cuda.current_device()}")
        self._dev_index = self.device.index
        if _compiled is None:
            self.cfg, self.tables, self.spec, self.param_names = compile_config(
                env, tunable_params, change_notification=change_notification,
                delta_change_notification=delta_change_notification, in_sim_change=in_sim_change,
                scalar_reward=scalar_reward, persistent_params=persistent_params, track_returns=track_returns,
                is_sim_env=is_sim_env, violation_mask=violation_mask, autoreset=autoreset, libm_exact=libm_exact, **kwargs)
        else:
            # a planning copy (fork): the SOURCE's compiled configuration with the copy's own flags and TimeLimit.  Nothing is
            # compiled again - sampled schedules (CustomScheduler, user-defined subclasses: ns_gym_amd.extension) must be the very
            # tables the source steps with, whatever has happened to the user's Python objects since the source was built
            from .envs import from_gym_env

            src_cfg, self.tables, _, self.param_names = _compiled
            self.spec = from_gym_env(env)
            self.cfg = A.Config.from_buffer_copy(bytes(src_cfg))
            self.cfg.flags = (src_cfg.flags & ~(A.F_SIM_ENV | A.F_IN_SIM_CHANGE)) | (A.F_SIM_ENV if is_sim_env else 0) | (A.F_IN_SIM_CHANGE if in_sim_change else 0)
            self.cfg.max_episode_steps = int(self.spec.max_episode_steps) if self.spec.max_episode_steps else 0
        self._ctor = dict(env=env, tunable_params=tunable_params, num_envs=num_envs, change_notification=change_notification,
                          delta_change_notification=delta_change_notification, in_sim_change=in_sim_change,
                          scalar_reward=scalar_reward, persistent_params=persistent_params, track_returns=track_returns,
                          device=device, violation_mask=violation_mask, specialize=specialize, autoreset=autoreset, libm_exact=libm_exact, **kwargs)
        self.tunable_params = tunable_params
        self.change_notification = change_notification
        self.delta_change_notification = delta_change_notification
        self.in_sim_change = in_sim_change
        self.scalar_reward = scalar_reward
        self.persistent_params = persistent_params
        self.autoreset = bool(autoreset)
        self.libm_exact = bool(self.cfg.flags & A.F_LIBM_EXACT)
        if self.libm_exact:
            specialize = True            # the exact arithmetic lives in the specialised units only
        self.frozen = False
        self.is_sim_env = bool(is_sim_env)
        self.has_reset = False
        self.num_envs = self.N = int(num_envs)
        self.is_grid = self.cfg.env_type in A.GRID_ENVS     # FrozenLake / CliffWalking / Bridge: int state, θ = slip distribution
        self.is_frozenlake = self.is_grid                  # historical alias used by the adaptors
        self.n_dist = A.N_DIST.get(self.cfg.env_type, 0)

        lay = A.Layout()
        _lib.check(self.lib.nsg_layout_query(C.byref(self.cfg), self.N, C.byref(lay)), "nsg_layout_query")
        self.layout = lay
        self.obs_dim = lay.obs_dim
        self.action_is_float = bool(lay.action_is_float)
        self.n_actions = lay.n_actions
        with torch.cuda.device(self.device):
            # every row is a view into ONE zeroed allocation (256-byte aligned rows; the counter shards last): a planning
            # copy costs one allocation instead of 25, and the N = 1 adaptors read a whole step back in one copy (host_rows)
            spans, off = {}, 0
            order = [f for f in A.BUFFER_FIELDS if f[0] != "counters"] + [f for f in A.BUFFER_FIELDS if f[0] == "counters"]
            for name, ct in order:
                n = getattr(lay, name)
                if name == "counters":
                    self._arena_head = off
                if n > 0:
                    spans[name] = (off, n * C.sizeof(ct), ct)
                    off += (n * C.sizeof(ct) + 255) & ~255
            self._arena = torch.zeros(max(off, 256), dtype=torch.uint8, device=self.device)
            self._spans = spans
            self.buf = {name: None for name, _ in A.BUFFER_FIELDS}
            for name, (o, nb, ct) in spans.items():
                self.buf[name] = self._arena[o:o + nb].view(_TORCH_DT[ct])
            self._bufs = A.Buffers(**{k: (v.data_ptr() if v is not None else None) for k, v in self.buf.items()})
            h = C.c_void_p()
            _lib.check(self.lib.nsg_create(C.byref(self.cfg), self.tables, len(self.tables), self.N, C.byref(h)),
                       "nsg_create")
            self._h = h
            _lib.check(self.lib.nsg_bind(self._h, C.byref(self._bufs)), "nsg_bind")
            if specialize or (specialize is None and self.N >= _AUTO_SPECIALIZE_MIN_ENVS):
                try:
                    self.specialize()
                except _lib.NsgError as e:
                    if specialize:   # asked for explicitly
                        raise
                    # a large batch on the precompiled generic kernels: correct, but 10-35 % slower than the numbers this
                    # package quotes (they read the configuration through scalar loads and spill SGPRs doing so)
                    warnings.warn(f"config-specialised kernels are not available ({e}); this batch of {self.N} envs runs on the "
                                  f"precompiled generic kernels, 10-35 % slower per step (libhiprtc.so is needed for nsg_specialize)",
                                  SpecializationUnavailableWarning, stacklevel=2)
        self._make_views()
        self._spaces = None   # built on first use: planning copies are made per simulation and rarely look at them
        self._zero_flags = None
        self._viol_seen = 0
        self._err_seen = [0, 0]

    def specialize(self):
        """Route this batch's step()/rollout() through kernels compiled for ITS configuration (nsg_specialize)."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nsg_specialize(self._h), "nsg_specialize")
        return self

    @property
    def has_user_defined_updates(self) -> bool:
        """Some tunable parameter is driven by a user-defined `UpdateFn` subclass (its chain is a sampled table: ns_gym_amd.extension)."""
        from . import extension

        return any(extension.is_user_update_fn(fn) for fn in self.tunable_params.values())

    @property
    def specialized(self) -> bool:
        return bool(self.lib.nsg_is_specialized(self._h))

    @property
    def kernels(self) -> str:
        """Which kernels `step()` / `rollout()` of this batch launch: "generic (precompiled)", or "config-specialised" plus where
        that unit came from - "(prebuilt)": shipped with the library, built and inspected when the library was
        (ns_gym_amd/prebuilt.py); "(hiprtc)": compiled by this process; "(disk cache)": compiled by an earlier process."""
        origin = int(self.lib.nsg_spec_origin(self._h))
        return {0: "generic (precompiled)", 1: "config-specialised (hiprtc)", 2: "config-specialised (disk cache)",
                3: "config-specialised (prebuilt)"}[origin]

    # ------------------------------------------------------------------ tensor views
    def _make_views(self):
        N, P, b = self.N, max(self.cfg.n_params, 1), self.buf
        self.t = b["t"]
        self.reward = b["reward"]
        self.terminated = b["terminated"].view(torch.bool)
        self.truncated = b["truncated"].view(torch.bool)
        self.gt_env_change = b["env_change"].view(P, N)[: self.cfg.n_params]
        self.gt_delta_change = b["delta_change"].view(P, N)[: self.cfg.n_params]
        # [P, N] 1 where this step's update was rejected by the physical-constraint checker (violation_mask=True)
        self.violation = b["violation"].view(P, N) if b["violation"] is not None else None
        rows = self.n_dist * P if self.is_grid else P
        # grid envs: param p occupies rows [p*n, (p+1)*n); no tunable params -> an empty [0, N] view
        self.theta = (b["theta"].view(rows, N) if b["theta"] is not None
                      else torch.zeros((0, N), dtype=torch.float64, device=self.device))
        if self.is_grid:
            self.state = b["cell"]
            self.prob = b["prob"]
            self._table_prob_blocked = (b["table_prob"].view(-1, self.n_dist, 256) if b["table_prob"] is not None else None)
        else:
            self.state = b["obs"].view(N, self.obs_dim)
            self._phys_blocked = b["phys"].view(-1, self.layout.phys_dim, 256)   # [chunk][F][256], see nsgym_hip.h

    @property
    def table_prob(self):
        """FrozenLake / CliffWalking: the probabilities baked into the wrapper's P table, dense [n, N] (gathered copy
        of the chunk-blocked rows)."""
        tb = self._table_prob_blocked
        return None if tb is None else tb.permute(1, 0, 2).reshape(self.n_dist, -1)[:, : self.N].contiguous()

    @property
    def phys(self):
        """Float64 integrator state as a dense [F, N] tensor (a gathered COPY: the device rows are chunk-blocked,
        [ceil(N/256)][F][256], so that a workgroup's state is one contiguous run)."""
        F = self.layout.phys_dim
        return self._phys_blocked.permute(1, 0, 2).reshape(F, -1)[:, : self.N].contiguous()

    def set_phys(self, dense):
        """Install a dense float64 [F, N] integrator state (the inverse of `phys`; the observation rows are NOT refreshed:
        meant for transition-level tests that overwrite the state between `reset` and `step`)."""
        F, nb = self.layout.phys_dim, self._phys_blocked.shape[0]
        full = torch.zeros((F, nb * 256), dtype=torch.float64, device=self.device)
        full[:, : self.N] = torch.as_tensor(dense, dtype=torch.float64, device=self.device).reshape(F, self.N)
        self._phys_blocked.copy_(full.view(F, nb, 256).permute(1, 0, 2))

    @property
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ gym-like API
    def reset(self, *, seed=None, options=None, mask=None):
        """`reset(seed=s)`: env i is seeded exactly like the reference's `reset(seed=s+i)` (or
        `seed[i]` for a sequence); `reset()` continues every stream (ns_gym/base.py:365-410)."""
        seeds, base = None, None
        if seed is not None:
            if np.isscalar(seed):
                base = int(seed)
                if base < 0:   # np.random.default_rng(-1) raises; a wrapped 2^64 - 1 would be a different stream, silently
                    raise ValueError(f"seed must be a non-negative integer (got {base})")
            else:
                s = np.asarray(seed, dtype=np.uint64)
                assert s.shape == (self.N,), "seed sequence must have one entry per env"
                if self.N == 1 or bool((np.diff(s.astype(np.int64, copy=False)) == 1).all()):
                    base = int(s[0])     # seed[i] = seed[0] + i: gymnasium's vector-env convention, given explicitly
                else:
                    seeds = torch.from_numpy(s.view(np.int64)).to(self.device)
            if base is not None and mask is not None:   # a masked re-seed cannot stay in the affine form
                seeds = torch.from_numpy(((np.arange(self.N, dtype=np.uint64) + np.uint64(base & (2**64 - 1)))).view(np.int64)).to(self.device)
                base = None
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        with torch.cuda.device(self.device):
            if base is not None:
                # env i <- base + i with nothing stored per env (classic-control envs re-derive their streams: nsgym_hip.h)
                _lib.check(self.lib.nsg_reset_seeded(self._h, C.c_uint64(base & (2**64 - 1)), self._stream), "nsg_reset_seeded")
            else:
                _lib.check(self.lib.nsg_reset(self._h, seeds.data_ptr() if seeds is not None else None,
                                              m.data_ptr() if m is not None else None, self._stream), "nsg_reset")
        self.has_reset = True
        return self._obs(), self._info()

    def step(self, actions):
        """One fused wrapper step for all envs.  `actions`: int32[N] (discrete) or float32[N] /
        float32[N,1] (continuous) device tensor."""
        act = self._as_actions(actions)
        if torch.cuda.current_device() == self._dev_index:   # the common case: no device-guard round trip
            rc = self.lib.nsg_step(self._h, act.data_ptr(), self._stream)
        else:
            with torch.cuda.device(self.device):
                rc = self.lib.nsg_step(self._h, act.data_ptr(), self._stream)
        if rc:
            _lib.check(rc, "nsg_step")
        return self._obs(), self.reward, self.terminated, self.truncated, self._info()

    def _step_raw(self, actions_ptr: int) -> None:
        """nsg_step on a raw pointer, nothing built or returned: the N = 1 adaptors hand over a pinned HOST word the kernel reads
        directly (unified addressing) and take the step's outputs from `host_rows()`."""
        if torch.cuda.current_device() == self._dev_index:
            rc = self.lib.nsg_step(self._h, actions_ptr, self._stream)
        else:
            with torch.cuda.device(self.device):
                rc = self.lib.nsg_step(self._h, actions_ptr, self._stream)
        if rc:
            _lib.check(rc, "nsg_step")

    def rollout(self, actions, record=("obs", "reward", "terminated", "truncated")):
        """K fused steps in one launch; `actions`: [K, N].  Returns a dict of [K, ...] trajectory
        tensors for the fields named in `record`."""
        K = int(actions.shape[0])
        dt = torch.float32 if self.action_is_float else torch.int32
        act = actions.to(device=self.device, dtype=dt).reshape(K, self.N).contiguous()
        P, N = max(self.cfg.n_params, 1), self.N
        shapes = {"obs": ((K, N) if self.is_frozenlake else (K, N, self.obs_dim),
                          torch.int32 if self.is_frozenlake else torch.float32),
                  "reward": ((K, N), torch.float32), "terminated": ((K, N), torch.uint8),
                  "truncated": ((K, N), torch.uint8), "env_change": ((K, P, N), torch.uint8),
                  "delta_change": ((K, P, N), torch.float32)}
        out = {k: torch.empty(shapes[k][0], dtype=shapes[k][1], device=self.device) for k in record}
        ro = A.RolloutOut(**{k: v.data_ptr() for k, v in out.items()})
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nsg_rollout(self._h, act.data_ptr(), K, C.byref(ro), self._stream), "nsg_rollout")
        for k in ("terminated", "truncated"):
            if k in out:
                out[k] = out[k].view(torch.bool)
        return out

    def rollout_policy(self, policy, k_steps: int, record=(), accounts=None, step0: int = 0, record_actions: bool = False):
        """K CLOSED-LOOP steps in one launch (`nsg_rollout_policy`): every env's action of step k is computed inside the kernel from
        what its step k - 1 produced - `policy` is a `ns_gym_amd.policies.Policy` (UniformRandom, TabularPolicy, LinearPolicy) -
        and `accounts` (`policies.EpisodeAccounts`) keeps each env's discounted return, length and alive flag in registers across
        the K steps.  With `record=()` nothing but the accounts leaves the launch.  `step0`: the index of this launch's first
        step in the caller's loop (UniformRandom's counter).  Returns the recorded [K, ...] tensors (+ "actions" when asked)."""
        K = int(k_steps)
        P, N = max(self.cfg.n_params, 1), self.N
        shapes = {"obs": ((K, N) if self.is_frozenlake else (K, N, self.obs_dim),
                          torch.int32 if self.is_frozenlake else torch.float32),
                  "reward": ((K, N), torch.float32), "terminated": ((K, N), torch.uint8),
                  "truncated": ((K, N), torch.uint8), "env_change": ((K, P, N), torch.uint8),
                  "delta_change": ((K, P, N), torch.float32)}
        out = {k: torch.empty(shapes[k][0], dtype=shapes[k][1], device=self.device) for k in record}
        ro = A.RolloutOut(**{k: v.data_ptr() for k, v in out.items()})
        acts = None
        if record_actions:
            acts = torch.empty((K, N), dtype=torch.float32 if self.action_is_float else torch.int32, device=self.device)
        pol = policy._struct(self, step0, acts)
        self._last_policy_kind = int(pol.kind)
        acc = accounts._struct() if accounts is not None else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nsg_rollout_policy(self._h, C.byref(pol), K, C.byref(ro), C.byref(acc) if acc is not None else None,
                                                   self._stream), "nsg_rollout_policy")
        for k in ("terminated", "truncated"):
            if k in out:
                out[k] = out[k].view(torch.bool)
        if acts is not None:
            out["actions"] = acts
        return out

    @property
    def policy_kernels(self) -> str:
        """Which kernel the last `rollout_policy` launched: the handle's specialised unit of that action source or the generic one."""
        return "config-specialised" if self.lib.nsg_rollout_policy_kind(self._h, int(getattr(self, "_last_policy_kind", 1))) == 1 else "generic"

    def _as_actions(self, actions):
        dt = torch.float32 if self.action_is_float else torch.int32
        if (type(actions) is torch.Tensor and actions.dtype == dt and actions.dim() == 1 and actions.shape[0] == self.N
                and actions.device == self.device and actions.is_contiguous()):
            return actions   # already what the kernel reads: no views, no copies
        act = torch.as_tensor(actions, device=self.device)
        if act.dtype != dt:
            act = act.to(dt)
        act = act.reshape(-1)
        assert act.numel() == self.N, f"expected {self.N} actions, got {act.numel()}"
        return act.contiguous()

    # ------------------------------------------------------------------ observation / info
    def _masked(self):
        hide = self.frozen or (self.is_sim_env and not self.in_sim_change)
        if self._zero_flags is None:
            self._zero_flags = (torch.zeros_like(self.gt_env_change), torch.zeros_like(self.gt_delta_change))
        ec = self.gt_env_change if (self.change_notification and not hide) else self._zero_flags[0]
        dc = self.gt_delta_change if (self.delta_change_notification and not hide) else self._zero_flags[1]
        return ec, dc

    def _rows(self, key, mat):
        """Per-param row views of a [P, N] buffer view, made once (the buffers never move)."""
        c = self._row_cache.get(key)
        if c is None or c[0] is not mat:
            c = (mat, {p: mat[j] for j, p in enumerate(self.param_names)})
            self._row_cache[key] = c
        return c[1]

    def _obs(self):
        """NS observation dict (ns_gym/base.py:343-348), batched: values are [N]-tensors per param."""
        ec, dc = self._masked()
        return {
            "state": self.state,
            "env_change": dict(self._rows("ec", ec)),
            "delta_change": dict(self._rows("dc", dc)),
            "relative_time": self.t,
        }

    def _info(self):
        info = {
            "Ground Truth Env Change": dict(self._rows("gt_ec", self.gt_env_change)),
            "Ground Truth Delta Change": dict(self._rows("gt_dc", self.gt_delta_change)),
        }
        if self.is_grid:
            info["prob"] = self.prob
            n = self.n_dist
            info["transition_prob"] = (self.theta if len(self.param_names) == 1 else
                                       {p: self.theta[j * n:(j + 1) * n] for j, p in enumerate(self.param_names)})
        return info

    def _get_spaces(self):
        """Per-env spaces (gymnasium.vector naming) and the NS observation Dict of base.py:275-292."""
        if self._spaces is None:
            from . import spaces

            state, action = spaces.base_spaces(self.spec.class_name, self.spec.desc)
            self._spaces = (state, action, spaces.ns_observation_space(state, self.param_names))
        return self._spaces

    single_state_space = property(lambda self: self._get_spaces()[0])
    single_action_space = property(lambda self: self._get_spaces()[1])
    single_observation_space = property(lambda self: self._get_spaces()[2])
    observation_space = property(lambda self: self._get_spaces()[2])
    action_space = property(lambda self: self._get_spaces()[1])

    def host_rows(self) -> dict:
        """Every row except the counter shards as NumPy arrays, fetched with ONE device-to-host copy (synchronises).
        Meant for small batches - the N = 1 adaptors read a step's outputs from it instead of one `.item()` per scalar."""
        st = self.__dict__.get("_host_stage")
        if st is None:
            # One pinned staging buffer per batch, reused by every call.  Up to 64 KB of rows (the N = 1 adaptors' case) come
            # over through nsg_read_back: a one-workgroup launch stores them into the pinned buffer and then publishes a
            # sequence number the host polls - no DMA copy, no event, no blocking wait (21 -> ~9 us per call).  Larger heads
            # use an asynchronous copy and a polled event.  The returned arrays are views of the buffer: valid until the next call.
            head = (self._arena_head + 15) & ~15
            buf = torch.zeros(head + 16, dtype=torch.uint8).pin_memory()
            st = self._host_stage = [buf, torch.cuda.Event(), buf.numpy()[head:head + 8].view(np.uint64), 0, head]
            self._host_views = {name: buf.numpy()[o:o + nb].view(_NP_DT[ct]) for name, (o, nb, ct) in self._spans.items()
                                if name != "counters"}
        buf, ev, flag, seq, head = st
        if head <= 65536 and head <= self._arena.numel():
            st[3] = seq = seq + 1
            with torch.cuda.device(self.device):
                _lib.check(self.lib.nsg_read_back(self._arena.data_ptr(), buf.data_ptr(), head, seq, self._stream), "nsg_read_back")
            spins = 0
            while int(flag[0]) != seq:
                spins += 1
                if spins == 2_000_000:        # ~1 s of polling: let the runtime report a fault instead of spinning for ever
                    torch.cuda.synchronize(self.device)
                    if int(flag[0]) != seq:
                        raise _lib.NsgError("nsg_read_back: the device finished without publishing the rows")
        else:
            buf[:self._arena_head].copy_(self._arena[:self._arena_head], non_blocking=True)
            ev.record(torch.cuda.current_stream(self.device))
            while not ev.query():
                pass
        return self._host_views

    # ------------------------------------------------------------------ reductions / bookkeeping
    def counters(self) -> dict:
        """Running totals produced by the kernels' wavefront ballots (synchronises)."""
        c = self.buf["counters"].view(A.CNT_COUNT, A.CNT_SHARDS).sum(dim=1).tolist()
        return {"episodes": int(c[A.CNT_DONE]), "updates_applied": int(c[A.CNT_FIRED]),
                "constraint_violations": int(c[A.CNT_VIOLATION]), "env_steps": int(c[A.CNT_STEPS]),
                "lc_exhausted": int(c[A.CNT_LC_EXHAUSTED]), "scheduler_overruns": int(c[A.CNT_SCHED_OVERRUN])}

    @property
    def may_raise(self) -> bool:
        """This configuration contains an update function / scheduler for which the reference can raise mid-run
        (LCBounded's rejection loop, a sampled CustomScheduler): the kernels count such events, `check_errors` raises."""
        return any(self.cfg.params[p].upd_kind == A.UPD_D_LCBOUNDED
                   or (self.cfg.params[p].sched_kind == A.SCHED_TABLE and self.cfg.params[p].sched_i0 == 2)
                   for p in range(self.cfg.n_params))

    def check_errors(self) -> None:
        """Raise what the reference would have raised inside `step()` (synchronises).  A kernel cannot raise, so it counts:
          * LCBoundedDistrubutionUpdate found no candidate within its Lipschitz bound in 1e5 tries -> ValueError
            (ns_gym/update_functions/distribution.py:168-182);
          * a CustomScheduler was asked about a t beyond the horizon its callable was sampled over (the reference would
            simply have called it, ns_gym/schedulers.py:31-43) -> ValueError naming the remedy.
        `step()` itself never synchronises; the N = 1 adaptors call this after every step, `run_episodes` at its end, a
        training loop whenever it reads results back."""
        c = self.counters()
        if c["lc_exhausted"] > self._err_seen[0]:
            n, self._err_seen[0] = c["lc_exhausted"] - self._err_seen[0], c["lc_exhausted"]
            L = [self.cfg.params[p].u[0] for p in range(self.cfg.n_params) if self.cfg.params[p].upd_kind == A.UPD_D_LCBOUNDED]
            raise ValueError(f"Could not find a Lipschitz-continuous update after {int(1e5)} attempts (L={L[0] if L else None}) "
                             f"in {n} (env, step) case(s); those distributions were left unchanged")
        if c["scheduler_overruns"] > self._err_seen[1]:
            n, self._err_seen[1] = c["scheduler_overruns"] - self._err_seen[1], c["scheduler_overruns"]
            raise ValueError(f"a CustomScheduler (or a Discrete / Window schedule cut at the reachable horizon) was asked {n} time(s) about a t beyond the horizon its event function was sampled "
                             f"over (it did not fire there); construct it with horizon=<largest t reached>")

    def check_constraints(self) -> int:
        """Aggregated ConstraintViolationWarning (the reference warns per violation,
        classic_control.py:212-234); returns the number of new violations since the last call."""
        v = self.counters()["constraint_violations"]
        new = v - self._viol_seen
        self._viol_seen = v
        if new > 0:
            warnings.warn(f"{new} parameter updates violated a physical constraint and were not applied",
                          ConstraintViolationWarning)
        return new

    def done_indices(self):
        """Dense int32 tensor of env indices whose episode ended in the last step (compaction of the
        wavefront ballot words; order unspecified)."""
        idx = torch.empty(self.N, dtype=torch.int32, device=self.device)
        cnt = torch.zeros(1, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nsg_compact_done(self._h, idx.data_ptr(), cnt.data_ptr(), self._stream), "nsg_compact_done")
        return idx[: int(cnt.item())]

    def episode_returns(self):
        """(return, length) of every env's last finished episode (length 0 = none yet).  CartPole pays +1 and MountainCar -1 on
        every step, so their return is +-length and no return row is kept for them (a derived tensor is returned)."""
        if self.buf["last_length"] is None:
            raise ValueError("construct with track_returns=True")
        length = self.buf["last_length"]
        if self.buf["last_return"] is not None:
            return self.buf["last_return"], length
        if getattr(self, "_ret_derived", None) is None:
            self._ret_derived = torch.empty(self.N, dtype=torch.float32, device=self.device)
        torch.mul(length, 1.0 if self.cfg.env_type == A.ENV_CARTPOLE else -1.0, out=self._ret_derived)   # one small kernel
        return self._ret_derived, length

    def time_steps(self, actions, iters: int) -> float:
        """Average device milliseconds per `nsg_step` launch over `iters` back-to-back launches,
        measured with hipEvents on the current stream."""
        act = self._as_actions(actions)
        ms = C.c_float()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nsg_time_steps(self._h, act.data_ptr(), int(iters), self._stream, C.byref(ms)),
                       "nsg_time_steps")
        return float(ms.value)

    # ------------------------------------------------------------------ planning copies
    def fork(self, theta_mode: int = 0, entropy: int | None = None, into: "VecNSEnv | None" = None,
             repeat: int = 1) -> "VecNSEnv":
        """Batched planning-env snapshot: a new `VecNSEnv` (is_sim_env=True) holding a copy of every
        env's state, t and update-fn state, with every stream re-seeded from `entropy` (fresh OS
        entropy by default, like the reference's `_reseed_planning_env_rngs`, ns_gym/base.py:433-441).
        theta_mode 0 keeps the current θ, 1 installs the construction-time θ.

        `into`: an earlier copy of THIS env to overwrite (a planner that snapshots once per simulation,
        MCTS.py:131): no allocation, no handle creation - one kernel launch.
        `repeat`: the copy holds `repeat` copies of every env (copy j <- env j mod N, each with its own
        streams): all simulations of one decision as ONE batch, advanced by one `rollout` launch."""
        import os

        if entropy is None:
            entropy = int.from_bytes(os.urandom(8), "little")
        if int(theta_mode) == 1 and self.in_sim_change and self.has_user_defined_updates:
            # get_planning_env() without delta notification hands the copy the INITIAL θ at the source's t, and with in_sim_change
            # the copy keeps calling `_update` from there: a chain that starts somewhere the sampled one never was
            raise _lib.NsgError("a planning copy that restarts from the initial θ (get_planning_env without delta_change_notification) and keeps "
                                "evolving (in_sim_change=True) leaves the θ chain a user-defined update function was sampled over: use "
                                "delta_change_notification=True (the copy continues the source's chain) or in_sim_change=False (frozen copy)")
        if into is not None:
            assert into.is_sim_env and into.N % self.N == 0 and getattr(into, "_fork_parent", None) is self._fork_root(), \
                "`into` must be a planning copy previously forked from this env"
            with torch.cuda.device(self.device):
                _lib.check(self.lib.nsg_fork(self._h, into._h, C.c_uint64(entropy & (2**64 - 1)), int(theta_mode), self._stream),
                           "nsg_fork")
            into.has_reset, into.frozen = self.has_reset, self.frozen
            return into
        kw = dict(self._ctor)
        kw["is_sim_env"] = True
        kw["device"] = self.device
        kw["num_envs"] = self.N * int(repeat)
        if self.spec.class_name in ("CliffWalkingEnv", "Bridge"):
            # the reference re-makes these copies with max_episode_steps=1000 (toy_text.py:229,685)
            import dataclasses

            kw["env"] = dataclasses.replace(self.spec, max_episode_steps=1000)
        dst = VecNSEnv(**kw, _compiled=(self.cfg, self.tables, self.spec, self.param_names))
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nsg_fork(self._h, dst._h, C.c_uint64(entropy & (2**64 - 1)), int(theta_mode), self._stream),
                       "nsg_fork")
        dst.has_reset = self.has_reset
        dst.frozen = self.frozen
        dst._fork_parent = self._fork_root()
        return dst

    def _fork_root(self):
        """The non-sim env a chain of copies descends from (copies of copies share the config family)."""
        return getattr(self, "_fork_parent", None) or self

    def get_planning_env(self) -> "VecNSEnv":
        """`get_planning_env()` of the reference wrappers for all envs at once
        (classic_control.py:120-136, toy_text.py:471-481): current θ if the agent is told the
        deltas (or this already is a planning copy), otherwise the initial θ; frozen unless
        in_sim_change."""
        assert self.has_reset, "The environment must be reset before getting the planning environment."
        keep = self.is_sim_env or self.delta_change_notification
        return self.fork(theta_mode=0 if keep else 1)

    def __deepcopy__(self, memo):
        """`copy.deepcopy(env)` -> planning copy with the current θ (classic_control.py:138-186)."""
        return self.fork(theta_mode=0)

    def seed_streams(self, seed, which: str = "env"):
        """Re-seed streams without touching env state: "env" = env.np_random, "update" = update fns."""
        s = (np.arange(self.N, dtype=np.uint64) + np.uint64(int(seed))) if np.isscalar(seed) else np.asarray(seed, dtype=np.uint64)
        assert s.shape == (self.N,)
        d = torch.from_numpy(s.view(np.int64)).to(self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nsg_seed_streams(self._h, d.data_ptr(), 0 if which == "env" else 1, self._stream),
                       "nsg_seed_streams")

    # ------------------------------------------------------------------ checkpoint / resume
    def _signature(self) -> str:
        import hashlib

        return hashlib.sha1(bytes(self.cfg) + bytes(self.tables) + str(self.N).encode()).hexdigest()

    def state_dict(self, to_cpu: bool = True) -> dict:
        """Checkpoint of the whole batch.  Every device row - state, θ, t, every PCG64 stream, list cursors, last outputs,
        counters - is a view into one allocation, so the device state IS that byte string; the host adds the two flags the
        reference keeps on the wrapper.  (The reference has no env checkpointing; its nearest mechanism, `deepcopy`,
        deliberately re-seeds the streams: base.py:433-441.)  A restored batch continues bit for bit."""
        arena = self._arena.cpu() if to_cpu else self._arena.clone()
        return {"arena": arena, "signature": self._signature(), "has_reset": self.has_reset, "frozen": self.frozen,
                "viol_seen": self._viol_seen, "err_seen": list(self._err_seen)}

    def load_state_dict(self, sd: dict):
        """Restore a `state_dict()` taken from a batch with the same configuration and size."""
        if sd.get("signature") != self._signature():
            raise ValueError("load_state_dict: the checkpoint was taken from a different configuration or batch size")
        if sd["arena"].numel() != self._arena.numel():
            raise ValueError("load_state_dict: buffer size mismatch")
        self._arena.copy_(sd["arena"].to(self.device, non_blocking=False))
        self.has_reset, self.frozen, self._viol_seen = bool(sd["has_reset"]), bool(sd["frozen"]), int(sd["viol_seen"])
        # the arena carries the raised-condition counters: what this object has already reported must follow them (a checkpoint
        # from before "err_seen" existed: re-baseline on the restored counters, i.e. nothing old is raised again)
        if "err_seen" in sd:
            self._err_seen = [int(x) for x in sd["err_seen"]]
        else:
            c = self.counters()
            self._err_seen = [c["lc_exhausted"], c["scheduler_overruns"]]
        return self

    def freeze(self, mode: bool = True):
        if not isinstance(mode, bool):
            raise TypeError(f"Expected mode to be a boolean, got {type(mode)}")
        self.frozen = mode
        return self

    def unfreeze(self):
        return self.freeze(False)

    def get_default_params(self):
        from .envs import TUNABLE_PARAMS

        return TUNABLE_PARAMS[self.spec.class_name]

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            torch.cuda.synchronize(self.device)
            self.lib.nsg_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ResidentStepper:
    """A closed loop in the launch-bound regime (`nsg_resident_start`, include/nsgym_hip.h): ONE kernel stays on the device and takes a
    step of `env` whenever the producer of the actions publishes the next row through the mailbox - no launch per step.

        loop = ResidentStepper(env, actions)            # actions: the int32 / float32 [N] device tensor the producer rewrites
        loop.start(max_steps)                           # zeroes the mailbox, launches the stepper on the loop's own HIGH-PRIORITY stream
        ... the producer: for k in 0 .. : wait for loop.step_seq >= k, write `actions`, publish act_seq = k + 1 ...
        status, steps = loop.result()                   # waits for the kernel to leave: "finished" / "starved" / "stopped"

    Every wait on the device is bounded (`wait_budget_us`, then a 200-us grace period): a producer that goes silent costs that long,
    never a hung process.  Leaving, the kernel writes every row back, so afterwards `env` is exactly where `steps` calls of `env.step`
    would have left it and ordinary `step()` / `rollout()` calls carry on.  Batches of at most 2^17 envs.

    Stepper and producer must RUN AT THE SAME TIME.  Two ordinary streams may share one hardware queue (the runtime multiplexes
    streams onto a few), in which case the second kernel would start only after the first has left - starved, safely, but useless.
    Streams of different priority never share a queue: the loop owns a high-priority stream for the stepper; give the producer an
    ordinary one."""

    MAX_CHUNKS = 512
    HEADER = 8                      # stop, status, steps_done, steps_max, 4 internal words; then act_seq[512], step_seq[512] (nsg_mailbox)
    WORDS = HEADER + 2 * MAX_CHUNKS
    STATUS = {0: "resident", 1: "finished", 2: "starved", 3: "stopped"}

    def __init__(self, env: "VecNSEnv", actions, wait_budget_us: int = 2000):
        self.env, self.budget = env, int(wait_budget_us)
        self.actions = env._as_actions(actions)
        self.mailbox = torch.zeros(self.WORDS, dtype=torch.int64, device=env.device)   # nsg_mailbox
        self.stream = torch.cuda.Stream(env.device, priority=-1)                       # the stepper's own (high-priority) stream
        self._stream = None

    def start(self, max_steps: int, stream=None, prefilled: int = 0):
        """`prefilled`: that many action rows count as published already (an open-loop run out of ONE action row: measurements)."""
        e = self.env
        cur = torch.cuda.current_stream(e.device)
        stream = stream or self.stream
        self.mailbox.zero_()                      # on the current stream; stepper and producer are ordered behind this point
        if prefilled:
            self.act_seq.fill_(int(prefilled))
        self._zeroed = torch.cuda.Event()
        self._zeroed.record(cur)
        stream.wait_event(self._zeroed)
        with torch.cuda.stream(stream):
            _lib.check(e.lib.nsg_resident_start(e._h, self.actions.data_ptr(), self.mailbox.data_ptr(), int(max_steps), self.budget,
                                                C.c_void_p(stream.cuda_stream)), "nsg_resident_start")
        self._stream = stream
        return self

    def demo_policy(self, max_steps: int, stream, watch: int = 2):
        """The library's stand-in producer (discrete-action classic-control envs), resident on `stream`: action[i] =
        ((obs[i][watch] > 0) + k) mod n_actions.  `stream` must differ from the stepper's; it is ordered behind the mailbox's zeroing."""
        e = self.env
        assert stream != self._stream, "stepper and producer must run concurrently: give the producer a stream of its own"
        stream.wait_event(self._zeroed)           # behind the mailbox's zeroing, NOT behind the stepper
        with torch.cuda.stream(stream):
            _lib.check(e.lib.nsg_resident_demo_policy(e._h, int(watch), self.actions.data_ptr(), self.mailbox.data_ptr(), int(max_steps), self.budget,
                                                      C.c_void_p(stream.cuda_stream)), "nsg_resident_demo_policy")
        return self

    def publish(self, step: int, stream=None):
        """Producer side for policies made of ordinary kernels: enqueue on the stream that has just written `actions` for `step`."""
        e = self.env
        stream = stream or torch.cuda.current_stream(e.device)
        _lib.check(e.lib.nsg_resident_publish(e._h, self.mailbox.data_ptr(), int(step), C.c_void_p(stream.cuda_stream)), "nsg_resident_publish")
        return self

    def stop(self, stream=None):
        """Raise `stop` from another stream: producer and stepper leave after at most one more step."""
        stream = stream or torch.cuda.Stream(self.env.device)
        with torch.cuda.stream(stream):
            self.mailbox[0:1].fill_(3)
        return self

    @property
    def act_seq(self):
        """int64[chunks] view: the producer's side of the hand-shake (chunk j's actions of step k in place -> k + 1)."""
        return self.mailbox[self.HEADER:self.HEADER + self.MAX_CHUNKS][: (self.env.N + 255) // 256]

    @property
    def step_seq(self):
        return self.mailbox[self.HEADER + self.MAX_CHUNKS:][: (self.env.N + 255) // 256]

    def result(self):
        """(status, steps) once the stepper has left (waits for it).  `steps`: what every chunk has taken - for "finished" always, for
        "stopped" / "starved" when the producer publishes for all chunks at once (`publish`); `self.steps_range` has (fewest, most)."""
        self._stream.synchronize()
        w = self.mailbox[: self.HEADER].cpu().tolist()
        self.steps_range = (int(w[2]), int(w[3]))
        return self.STATUS.get(int(w[1]), str(w[1])), int(w[2])


def step_group(envs, actions):
    """One heterogeneous launch over several VecNSEnv of different env types (per-env-type
    dispatch is uniform per workgroup)."""
    lib = _lib.load()
    n = len(envs)
    acts = [e._as_actions(a) for e, a in zip(envs, actions)]
    hs = (C.c_void_p * n)(*[e._h for e in envs])
    ap = (C.c_void_p * n)(*[a.data_ptr() for a in acts])
    with torch.cuda.device(envs[0].device):
        _lib.check(lib.nsg_step_group(hs, n, ap, envs[0]._stream), "nsg_step_group")
    return [(e._obs(), e.reward, e.terminated, e.truncated, e._info()) for e in envs]


def rollout_group(envs, actions, record=("obs", "reward", "terminated", "truncated")):
    """K fused steps of several VecNSEnv of different env types in ONE launch (`nsg_rollout_group`): `actions[k]` is member k's
    [K, N_k] tensor.  Returns one dict of [K, ...] trajectory tensors per member, like `VecNSEnv.rollout`."""
    lib = _lib.load()
    n = len(envs)
    K = int(actions[0].shape[0])
    acts, outs = [], []
    ros = (A.RolloutOut * n)()
    for j, (e, a) in enumerate(zip(envs, actions)):
        assert int(a.shape[0]) == K, "every member takes the same number of steps"
        dt = torch.float32 if e.action_is_float else torch.int32
        acts.append(a.to(device=e.device, dtype=dt).reshape(K, e.N).contiguous())
        P, N = max(e.cfg.n_params, 1), e.N
        shapes = {"obs": ((K, N) if e.is_grid else (K, N, e.obs_dim), torch.int32 if e.is_grid else torch.float32),
                  "reward": ((K, N), torch.float32), "terminated": ((K, N), torch.uint8), "truncated": ((K, N), torch.uint8),
                  "env_change": ((K, P, N), torch.uint8), "delta_change": ((K, P, N), torch.float32)}
        out = {k: torch.empty(shapes[k][0], dtype=shapes[k][1], device=e.device) for k in record}
        ros[j] = A.RolloutOut(**{k: v.data_ptr() for k, v in out.items()})
        outs.append(out)
    hs = (C.c_void_p * n)(*[e._h for e in envs])
    ap = (C.c_void_p * n)(*[a.data_ptr() for a in acts])
    with torch.cuda.device(envs[0].device):
        _lib.check(lib.nsg_rollout_group(hs, n, ap, K, ros, envs[0]._stream), "nsg_rollout_group")
    for out in outs:
        for k in ("terminated", "truncated"):
            if k in out:
                out[k] = out[k].view(torch.bool)
    return outs


GROUP_KINDS = {0: "unplanned", 1: "generic", 2: "generic-full", 3: "specialised", 4: "specialised (prebuilt)"}


def step_group_kind(envs) -> str:
    """Which kernel the current plan of this member list launches (`nsg_step_group_kind`): "unplanned" before the first
    `step_group` of the list (or after a member changed), "generic" / "generic-full" (precompiled kernels), "specialised"
    (the unit compiled for the ordered tuple of the members' configurations)."""
    lib = _lib.load()
    n = len(envs)
    hs = (C.c_void_p * n)(*[e._h for e in envs])
    k = lib.nsg_step_group_kind(hs, n)
    if k < 0:
        _lib.check(k, "nsg_step_group_kind")
    return GROUP_KINDS[k]
