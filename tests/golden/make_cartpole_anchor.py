#!/usr/bin/env python3
"""Generate tests/golden/cartpole_intree_anchor.npz from the ONE integrator formula the reference tree itself holds:
`NSCartPoleV0.transition` (ns_gym/benchmark_algorithms/rats-experiments/code/envs/nscartpole_v0.py:84-115).

Runs only in the build container (needs /root/reference, read-only); nothing is copied, the reference class is imported
and driven, and NUMBERS are stored.  The file imports the legacy `gym` package, which is not installed (ordinary
ModuleNotFoundError): the names it needs at import / construction time (`gym.Env`, `spaces.Discrete`, `spaces.Box`,
`seeding.np_random`) are bound to inert stand-ins; none of them takes part in `transition`.

Why this pins CartPole-v1's Euler step: with `time = 0` the inclination term vanishes exactly
(`alpha = alpha_max * sin(0) = 0`, `cos(alpha) = 1.0`, `gravity * sin(alpha) = 0.0`), the five-action force map gives
`-force_mag` for action 0 and `+force_mag` for action 4 (`-10 + 4 * 2 * 10 / 4`), and with `is_model_dynamic=False`
the time component stays 0.  What remains (:98-108) is the force / temp / thetaacc / xacc block and the Euler update of
gymnasium's CartPoleEnv with the same constants (:24-36).  It is NOT the same floating-point expression everywhere:
the legacy file multiplies left to right (`polemass_length * theta_dot * theta_dot * sintheta`,
`masspole * costheta * costheta / total_mass`) where gymnasium 1.2.1 squares first (`np.square(theta_dot)`,
`np.square(costheta)`), and it calls `math.sin/cos` where gymnasium calls NumPy's.  The anchor therefore agrees with
the restated CartPole-v1 step to a few float64 ulps (not bit for bit), and with `done` except on a threshold tie;
tests/test_cartpole_intree_anchor.py states the allowance.  Reward is not comparable (legacy: 0.0 on the terminating
step, gymnasium: 1.0) and is not recorded as an expectation.

Usage:  python tests/golden/make_cartpole_anchor.py
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("NSG_REFERENCE", "/root/reference")
SRC = os.path.join(REFERENCE, "ns_gym", "benchmark_algorithms", "rats-experiments", "code", "envs", "nscartpole_v0.py")

# θ settings: (gravity, masscart, masspole, force_mag, tau, length) in TUNABLE_PARAMS order (ns_gym/base.py:611-618).
# Row 0 = the defaults (nscartpole_v0.py:24-33 == CartPole-v1's); the others are what the wrapper's update functions
# produce in the BASELINE configs (masspole grown by IncrementUpdate, gravity moved by a RandomWalk) and one of each kind.
DEFAULTS = (9.8, 1.0, 0.1, 10.0, 0.02, 0.5)
DELTAS = [
    (0.0, 0.0, 0.0, 0.0, 0.0, 0.0),
    (0.0, 0.0, 0.1, 0.0, 0.0, 0.0),
    (0.0, 0.0, 0.5, 0.0, 0.0, 0.0),        # the reference tests' own case: total_mass 1.6, polemass_length 0.3
    (0.0, 0.0, 4.9, 0.0, 0.0, 0.0),
    (1.2544943667397455, 0.0, 0.0, 0.0, 0.0, 0.0),
    (-3.7, 0.0, 0.0, 0.0, 0.0, 0.0),
    (0.0, 0.75, 0.0, 0.0, 0.0, 0.0),
    (0.0, 0.0, 0.0, 5.5, 0.0, 0.0),
    (0.0, 0.0, 0.0, 0.0, 0.03, 0.0),
    (0.0, 0.0, 0.0, 0.0, 0.0, 0.45),
    (2.5, 0.3, 0.25, -2.0, 0.005, 0.2),
    (-9.8, 2.0, 1.9, 20.0, -0.01, 1.5),    # gravity exactly 0 is allowed by the wrapper's checker (rejects < 0 only)
]


def load_reference_class():
    gym = types.ModuleType("gym")

    class Env:
        pass

    gym.Env = Env
    spaces = types.ModuleType("gym.spaces")
    spaces.Discrete = lambda n: ("Discrete", n)
    spaces.Box = lambda low, high: ("Box", low, high)
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")
    seeding.np_random = lambda seed=None: (np.random.default_rng(seed), seed)
    utils.seeding = seeding
    gym.spaces, gym.utils = spaces, utils
    sys.modules.update({"gym": gym, "gym.spaces": spaces, "gym.utils": utils, "gym.utils.seeding": seeding})
    spec = importlib.util.spec_from_file_location("_ref_nscartpole_v0", SRC)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.NSCartPoleV0


def main():
    cls = load_reference_class()
    rng = np.random.default_rng(20260402)
    m = 512
    # states: half reset-like (the envelope CartPole-v1 episodes live in), half across and beyond the termination
    # thresholds (|x| > 2.4, |theta| > 0.2095) with fast poles
    lo = np.array([-0.05, -0.05, -0.05, -0.05]); hi = -lo
    a = rng.uniform(lo, hi, size=(m // 4, 4))
    b = rng.uniform([-2.3, -2.5, -0.2, -3.0], [2.3, 2.5, 0.2, 3.0], size=(m // 4, 4))
    c = rng.uniform([-2.6, -3.0, -0.26, -3.5], [2.6, 3.0, 0.26, 3.5], size=(m // 2, 4))
    states = np.concatenate([a, b, c]).astype(np.float64)
    actions = rng.integers(2, size=m).astype(np.int32)          # CartPole-v1 action; legacy action = 4 * action
    thetas = np.array([[d0 + dk for d0, dk in zip(DEFAULTS, row)] for row in DELTAS], dtype=np.float64)
    nxt = np.zeros((len(thetas), m, 4), dtype=np.float64)
    done = np.zeros((len(thetas), m), dtype=np.uint8)
    for s, th in enumerate(thetas):
        env = cls()
        env.gravity, env.masscart, env.masspole, env.force_mag, env.tau, env.length = (float(v) for v in th)
        # the wrapper's _dependency_resolver (ns_gym/wrappers/classic_control.py:426-444) keeps these two consistent
        env.total_mass = env.masspole + env.masscart
        env.polemass_length = env.length * env.masspole
        for i in range(m):
            st = tuple(float(v) for v in states[i]) + (0.0,)
            sp, _reward, d = env.transition(st, 4 * int(actions[i]), False)
            assert sp[4] == 0.0
            nxt[s, i] = sp[:4]
            done[s, i] = 1 if d else 0
    out = os.path.join(HERE, "cartpole_intree_anchor.npz")
    np.savez_compressed(out, states=states, actions=actions, thetas=thetas, deltas=np.array(DELTAS, dtype=np.float64),
                        next_state=nxt, done=done)
    print(f"wrote {out}: {len(thetas)} theta settings x {m} states, {int(done.sum())} terminating transitions")


if __name__ == "__main__":
    main()
