"""BASELINE.json's configurations as named workloads ("Configs" + §8(d) "Synthetic inputs"): what
`bench.py`, `tools/kbench.py` and the full-size parity tests construct.  Each entry: env id, tunable_params factory,
`make()` kwargs, wrapper kwargs, algorithmic bytes per env-step of the step API with float64 internal state,
and the batch size BASELINE quotes the configuration at."""
from __future__ import annotations

from . import make
from .schedulers import ContinuousScheduler, DiscreteScheduler, PeriodicScheduler
from .update_functions import DistributionStepWiseUpdate, IncrementUpdate, RandomWalk

WORKLOADS = {
    # C1 / C5: CartPole masspole IncrementUpdate(+0.1) via ContinuousScheduler
    "c1": dict(env_id="CartPole-v1", params=lambda: {"masspole": IncrementUpdate(ContinuousScheduler(), k=0.1)},
               make_kwargs={}, wrapper_kwargs={}, bytes_per_env_step=120, baseline_envs=1 << 20),
    # C2: CartPole gravity RandomWalk via PeriodicScheduler(period=3)
    "c2": dict(env_id="CartPole-v1", params=lambda: {"gravity": RandomWalk(PeriodicScheduler(period=3))},
               make_kwargs={}, wrapper_kwargs={}, bytes_per_env_step=157, baseline_envs=65536),
    # C3: FrozenLake 8x8, slip distribution stepped to [0.6, 0.2, 0.2] at t = 50
    "c3": dict(env_id="FrozenLake-v1",
               params=lambda: {"P": DistributionStepWiseUpdate(DiscreteScheduler({50}), [[0.6, 0.2, 0.2]])},
               make_kwargs={"map_name": "8x8"}, wrapper_kwargs={"initial_prob_dist": [1.0, 0.0, 0.0]},
               bytes_per_env_step=96, baseline_envs=1 << 20),
    # C4: Pendulum + Acrobot, 262 144 each, one heterogeneous launch
    "pend": dict(env_id="Pendulum-v1", params=lambda: {"m": IncrementUpdate(ContinuousScheduler(), k=0.01)},
                 make_kwargs={}, wrapper_kwargs={}, bytes_per_env_step=83, baseline_envs=1 << 18),
    "acro": dict(env_id="Acrobot-v1", params=lambda: {"LINK_MASS_2": IncrementUpdate(ContinuousScheduler(), k=0.1)},
                 make_kwargs={}, wrapper_kwargs={}, bytes_per_env_step=127, baseline_envs=1 << 18),
    # not BASELINE configs; priced the same way for the tables of DESIGN.md
    "mcar": dict(env_id="MountainCar-v0", params=lambda: {"force": IncrementUpdate(ContinuousScheduler(), k=1e-6)},
                 make_kwargs={}, wrapper_kwargs={}, bytes_per_env_step=79, baseline_envs=1 << 20),
    "mcarc": dict(env_id="MountainCarContinuous-v0", params=lambda: {"power": IncrementUpdate(ContinuousScheduler(), k=1e-6)},
                  make_kwargs={}, wrapper_kwargs={}, bytes_per_env_step=79, baseline_envs=1 << 20),
}


def build(name: str, n: int | None = None, *, track_returns: bool = True, specialize=None, device=None, seed=0, **extra):
    """A reset `VecNSEnv` of the named workload with `n` envs (default: the size BASELINE quotes it at)."""
    from .vec_env import VecNSEnv

    w = WORKLOADS[name]
    kw = dict(change_notification=True, delta_change_notification=True, track_returns=track_returns, specialize=specialize,
              device=device, **w["wrapper_kwargs"], **extra)
    env = VecNSEnv(make(w["env_id"], **w["make_kwargs"]), w["params"](), int(n or w["baseline_envs"]), **kw)
    if seed is not None:
        env.reset(seed=seed)
    return env


def random_actions(env, generator=None):
    """One synthetic random action per env, resident on the env's device (uniform over the action set,
    Pendulum U(-2, 2))."""
    import torch

    if env.action_is_float:
        return torch.rand(env.N, device=env.device, generator=generator) * 4 - 2
    return torch.randint(0, env.n_actions, (env.N,), dtype=torch.int32, device=env.device, generator=generator)
