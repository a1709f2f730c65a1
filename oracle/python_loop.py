"""TEST INFRASTRUCTURE (CPU baseline leg of bench.py, tests) - never imported by the product.

The reference's hot path as it executes today: one Python wrapper OBJECT per env instance, one `step()` call
per env per time step, dict observations, per-parameter scheduler / update-function calls.  Restated op for op
for the C1 / C2 configurations over the restated gymnasium base env (oracle/gym_restatement.py):

    NSClassicControlWrapper.step   ns_gym/wrappers/classic_control.py:60-100
      -> UpdateFn.__call__ / Scheduler.__call__      ns_gym/base.py:124-149, 67-81
      -> _constraint_checker (CartPole)             classic_control.py:208-235
      -> _dependency_resolver                        classic_control.py:426-444
      -> NSWrapper.step                              ns_gym/base.py:296-363
      -> gymnasium TimeLimit(CartPoleEnv).step       [UPSTREAM]

This is what BASELINE.md §3 calls the "object-model loop, 1 core": the figure the reference itself would show
on the same host, since its own package cannot travel to the GPU box.  tests/test_oracle_python_loop.py checks
it against the C oracle."""
from __future__ import annotations

import copy

import numpy as np

from . import gym_restatement as G


class _Scheduler:
    def __init__(self, start=0, end=np.inf):
        self.start, self.end = start, end

    def __call__(self, t):                      # base.py:67-81
        if self.start <= t <= self.end:
            return self._check(t)
        return False


class ContinuousScheduler(_Scheduler):
    def _check(self, t):                        # schedulers.py:52-53
        return True


class PeriodicScheduler(_Scheduler):
    def __init__(self, period, start=0, end=np.inf):
        super().__init__(start, end)
        self.period = period

    def _check(self, t):                        # schedulers.py:88-89
        return t % self.period == 0


class _UpdateFn:
    def __init__(self, scheduler):
        self.scheduler = scheduler
        self.prev_param, self.prev_time = None, -1

    def __call__(self, param, t):               # base.py:124-149
        assert isinstance(t, (int, float))
        if self.scheduler(t):
            updated = self._update(copy.copy(param), t)
            delta = updated - param             # base.py:182
            self.prev_param, self.prev_time = param, t
            return updated, 1, delta
        self.prev_param, self.prev_time = param, t
        return param, 0, 0.0


class IncrementUpdate(_UpdateFn):
    def __init__(self, scheduler, k):
        super().__init__(scheduler)
        self.k = k

    def _update(self, param, t):                # single_param.py:173-175
        param += self.k
        return param


class RandomWalk(_UpdateFn):
    def __init__(self, scheduler, mu=0, sigma=1, seed=None):
        super().__init__(scheduler)
        self.mu, self.sigma = mu, sigma
        self.rng = np.random.default_rng(seed)

    def seed(self, seed):                       # base.py:151-158
        self.rng = np.random.default_rng(seed)

    def _update(self, param, t):                # single_param.py:110-113
        return param + self.rng.normal(self.mu, self.sigma)


class PyNSCartPole:
    """One non-stationary CartPole env object (the reference builds one of these per env instance)."""

    def __init__(self, tunable_params, change_notification=False, delta_change_notification=False):
        self.env = G.make("CartPole-v1")
        self.unwrapped = self.env.unwrapped
        self.tunable_params = tunable_params
        self.init_initial_params = copy.deepcopy(tunable_params)
        self.initial_values = {p: getattr(self.unwrapped, p) for p in tunable_params}
        self.change_notification, self.delta_change_notification = change_notification, delta_change_notification
        self.t = 0

    def reset(self, *, seed=None):              # base.py:365-410 + classic_control.py:102-109
        state, info = self.env.reset(seed=seed)
        self.t = 0
        old = self.tunable_params
        self.tunable_params = copy.deepcopy(self.init_initial_params)
        if seed is not None:
            children = np.random.SeedSequence(seed).spawn(len(self.tunable_params))
            for child, fn in zip(children, self.tunable_params.values()):
                if hasattr(fn, "seed"):
                    fn.seed(child)
        else:
            for k, fn in self.tunable_params.items():
                if hasattr(old[k], "rng"):
                    fn.rng = old[k].rng
        for p, v in self.initial_values.items():
            setattr(self.unwrapped, p, copy.deepcopy(v))
        self._dependency_resolver()
        zeros = {p: 0 for p in self.tunable_params}
        return {"state": state, "env_change": zeros, "delta_change": dict(zeros), "relative_time": self.t}, info

    def _constraint_checker(self, new_vals):    # classic_control.py:208-235 (CartPole)
        out = {}
        for k, v in new_vals.items():
            out[k] = (v < 0) if k == "gravity" else (v <= 0) if k in ("masscart", "masspole", "length") else False
        return out

    def _dependency_resolver(self):             # classic_control.py:426-444
        u = self.unwrapped
        u.total_mass = u.masspole + u.masscart
        u.polemass_length = u.length * u.masspole

    def step(self, action):                     # classic_control.py:60-100 -> base.py:296-363
        env_change, delta_change, new_vals = {}, {}, {}
        for p, fn in self.tunable_params.items():
            cur = getattr(self.unwrapped, p)
            new, flag, delta = fn(cur, self.t)
            delta_change[p], env_change[p], new_vals[p] = delta, flag, new
        for k, violated in self._constraint_checker(new_vals).items():
            if not violated:
                setattr(self.unwrapped, k, new_vals[k])
            else:
                delta_change[k], env_change[k] = 0.0, 0
        self._dependency_resolver()
        state, reward, terminated, truncated, info = self.env.step(action)
        self.t += 1
        default_ec = {p: 0 for p in self.tunable_params}
        default_dc = {p: 0.0 for p in self.tunable_params}
        calc_ec = {k: int(v) for k, v in env_change.items()}
        calc_dc = {k: float(v) for k, v in delta_change.items()}
        obs = {"state": state,
               "env_change": {**default_ec, **calc_ec} if self.change_notification else default_ec,
               "delta_change": {**default_dc, **calc_dc} if self.delta_change_notification else default_dc,
               "relative_time": self.t}
        info["Ground Truth Env Change"], info["Ground Truth Delta Change"] = calc_ec, calc_dc
        info["prob"] = 1.0
        return obs, reward, terminated, truncated, info


def make_c1():
    return PyNSCartPole({"masspole": IncrementUpdate(ContinuousScheduler(), k=0.1)}, True, True)


def make_c2():
    return PyNSCartPole({"gravity": RandomWalk(PeriodicScheduler(period=3))}, True, True)


def run_loop(envs, actions, steps):
    """The caller's loop (evaluate/run_experiment.py:108-129 shape): step every env object, reset the finished ones."""
    n = len(envs)
    done = [False] * n
    total = 0
    for k in range(steps):
        a = actions[k % len(actions)]
        for i, env in enumerate(envs):
            if done[i]:
                env.reset()
                done[i] = False
            else:
                _, _, term, trunc, _ = env.step(int(a[i]))
                done[i] = term or trunc
            total += 1
    return total


if __name__ == "__main__":   # one worker of bench.py's all-cores leg: prints its own env-steps per second
    import sys
    import time

    import numpy as np

    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    envs = [make_c1() for _ in range(64)]
    for i, e in enumerate(envs):
        e.reset(seed=i)
    acts = np.random.default_rng(123).integers(2, size=(8, 64))
    run_loop(envs, acts, 20)
    t0 = time.perf_counter()
    n = run_loop(envs, acts, steps)
    print(n / (time.perf_counter() - t0))

