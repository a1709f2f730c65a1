"""TEST INFRASTRUCTURE ONLY — CPU restatement of the gymnasium 1.2.1 base MDPs.

The reference (ns_gym) does not contain the transition dynamics it perturbs: its
wrappers mutate attributes on a *gymnasium* env and then call gymnasium's ``step``
(`ns_gym/base.py:313`, `ns_gym/base.py:377`, `ns_gym/wrappers/classic_control.py:81,89`,
`ns_gym/wrappers/toy_text.py:339,367,398`).  gymnasium is pinned at 1.2.1
(`uv.lock:958-959`) and is NOT installed in this image (no network), so this module
restates its published algorithm for the envs on the hot path:

  CartPoleEnv, PendulumEnv, AcrobotEnv, MountainCarEnv, Continuous_MountainCarEnv,
  FrozenLakeEnv (+ TimeLimit, ``utils.seeding.np_random``, ``categorical_sample``).

Status: **integrator arithmetic = parity unpinned** [UPSTREAM, restated from the
published gymnasium sources].  The only in-tree corroboration is the legacy CartPole
block at `ns_gym/benchmark_algorithms/rats-experiments/code/envs/nscartpole_v0.py:24-36,92-108`
(same constants, same Euler formula).  Everything that the reference itself owns
(schedulers, update functions, wrapper ordering/masking/constraints/reset/seeding)
is pinned separately by golden vectors generated from the reference's own classes
running on top of THESE env classes (see ``tests/golden/make_golden.py``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may
import this module.  Nothing under ``ns_gym_amd/`` does.

The classes deliberately keep gymnasium's attribute names (``gravity``, ``masscart``,
``total_mass``, ``polemass_length``, ``LINK_MASS_2``, ``P``, ``s`` ...) because the
reference wrapper reads/writes them by name, and gymnasium's class names because the
reference dispatches on ``unwrapped.__class__.__name__``.
"""
from __future__ import annotations

import math
from typing import Any

import numpy as np

# --------------------------------------------------------------------------- spaces


class Space:
    def __init__(self, seed=None):
        self._np_random = None
        self._seed = seed

    @property
    def np_random(self):
        if self._np_random is None:
            self._np_random = np.random.default_rng(self._seed)
        return self._np_random

    def seed(self, seed=None):
        self._np_random = np.random.default_rng(seed)


class Discrete(Space):
    def __init__(self, n, seed=None, start=0):
        super().__init__(seed)
        self.n = int(n)
        self.start = int(start)
        self.shape = ()
        self.dtype = np.int64

    def sample(self):
        return int(self.start + self.np_random.integers(self.n))

    def contains(self, x):
        return isinstance(x, (int, np.integer)) and self.start <= int(x) < self.start + self.n

    def __eq__(self, o):
        return isinstance(o, Discrete) and o.n == self.n and o.start == self.start


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        super().__init__(seed)
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.shape(low)
        self.shape = tuple(shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=np.float64), self.shape).astype(self.dtype)
        self.high = np.broadcast_to(np.asarray(high, dtype=np.float64), self.shape).astype(self.dtype)

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return self.np_random.uniform(lo, hi, size=self.shape).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class Dict(Space):
    def __init__(self, spaces=None, seed=None):
        super().__init__(seed)
        self.spaces = dict(spaces or {})

    def __getitem__(self, k):
        return self.spaces[k]

    def keys(self):
        return self.spaces.keys()

    def sample(self):
        return {k: s.sample() for k, s in self.spaces.items()}


# --------------------------------------------------------------------------- core


def np_random(seed=None):
    """gymnasium.utils.seeding.np_random [UPSTREAM]: Generator(PCG64(SeedSequence(seed)))."""
    seed_seq = np.random.SeedSequence(seed)
    return np.random.Generator(np.random.PCG64(seed_seq)), seed_seq.entropy


class Env:
    metadata: dict = {}
    render_mode = None
    spec = None
    _np_random = None
    _np_random_seed = None

    @property
    def unwrapped(self):
        return self

    @property
    def np_random(self):
        if self._np_random is None:
            self._np_random, self._np_random_seed = np_random()
        return self._np_random

    @np_random.setter
    def np_random(self, value):
        self._np_random = value

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self._np_random, self._np_random_seed = np_random(seed)

    def step(self, action):
        raise NotImplementedError

    def close(self):
        pass


class Wrapper(Env):
    """gymnasium.Wrapper 1.x: no generic attribute forwarding; only the named properties."""

    def __init__(self, env):
        self.env = env
        self._action_space = None
        self._observation_space = None

    @property
    def unwrapped(self):
        return self.env.unwrapped

    @property
    def action_space(self):
        return self.env.action_space if self._action_space is None else self._action_space

    @action_space.setter
    def action_space(self, s):
        self._action_space = s

    @property
    def observation_space(self):
        return self.env.observation_space if self._observation_space is None else self._observation_space

    @observation_space.setter
    def observation_space(self, s):
        self._observation_space = s

    @property
    def spec(self):
        return self.env.spec

    @property
    def np_random(self):
        return self.env.np_random

    def step(self, action):
        return self.env.step(action)

    def reset(self, *, seed=None, options=None):
        return self.env.reset(seed=seed, options=options)

    def close(self):
        return self.env.close()

    def __repr__(self):
        return f"<{type(self).__name__}{self.env!r}>"


class TimeLimit(Wrapper):
    """gymnasium.wrappers.TimeLimit [UPSTREAM]: truncated when elapsed >= max_episode_steps."""

    def __init__(self, env, max_episode_steps):
        super().__init__(env)
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = None

    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(action)
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            truncated = True
        return obs, reward, terminated, truncated, info

    def reset(self, *, seed=None, options=None):
        self._elapsed_steps = 0
        return self.env.reset(seed=seed, options=options)


class EnvSpec:
    def __init__(self, id, max_episode_steps=None, kwargs=None):
        self.id = id
        self.max_episode_steps = max_episode_steps
        self.kwargs = dict(kwargs or {})


# --------------------------------------------------------------------------- CartPole


class CartPoleEnv(Env):
    """gymnasium.envs.classic_control.cartpole.CartPoleEnv [UPSTREAM 1.2.1], euler integrator.

    In-tree corroboration: rats-experiments/code/envs/nscartpole_v0.py:24-36 (constants),
    :92-108 (same force/temp/thetaacc/xacc block and Euler update).
    """

    def __init__(self, sutton_barto_reward=False, render_mode=None):
        self._sutton_barto_reward = sutton_barto_reward
        self.gravity = 9.8
        self.masscart = 1.0
        self.masspole = 0.1
        self.total_mass = self.masspole + self.masscart
        self.length = 0.5  # half the pole's length
        self.polemass_length = self.masspole * self.length
        self.force_mag = 10.0
        self.tau = 0.02
        self.kinematics_integrator = "euler"
        self.theta_threshold_radians = 12 * 2 * math.pi / 360
        self.x_threshold = 2.4
        high = np.array(
            [self.x_threshold * 2, np.inf, self.theta_threshold_radians * 2, np.inf],
            dtype=np.float32,
        )
        self.action_space = Discrete(2)
        self.observation_space = Box(-high, high, dtype=np.float32)
        self.render_mode = render_mode
        self.state = None
        self.steps_beyond_terminated = None

    def step(self, action):
        assert self.state is not None, "Call reset before using step method."
        x, x_dot, theta, theta_dot = self.state
        force = self.force_mag if action == 1 else -self.force_mag
        costheta = np.cos(theta)
        sintheta = np.sin(theta)
        temp = (force + self.polemass_length * np.square(theta_dot) * sintheta) / self.total_mass
        thetaacc = (self.gravity * sintheta - costheta * temp) / (
            self.length * (4.0 / 3.0 - self.masspole * np.square(costheta) / self.total_mass)
        )
        xacc = temp - self.polemass_length * thetaacc * costheta / self.total_mass
        x = x + self.tau * x_dot
        x_dot = x_dot + self.tau * xacc
        theta = theta + self.tau * theta_dot
        theta_dot = theta_dot + self.tau * thetaacc
        self.state = np.array((x, x_dot, theta, theta_dot), dtype=np.float64)
        terminated = bool(
            x < -self.x_threshold
            or x > self.x_threshold
            or theta < -self.theta_threshold_radians
            or theta > self.theta_threshold_radians
        )
        if not terminated:
            reward = 0.0 if self._sutton_barto_reward else 1.0
        elif self.steps_beyond_terminated is None:
            self.steps_beyond_terminated = 0
            reward = -1.0 if self._sutton_barto_reward else 1.0
        else:
            self.steps_beyond_terminated += 1
            reward = -1.0 if self._sutton_barto_reward else 0.0
        return np.array(self.state, dtype=np.float32), reward, terminated, False, {}

    def reset(self, *, seed=None, options=None):
        super().reset(seed=seed)
        low, high = -0.05, 0.05
        self.state = self.np_random.uniform(low=low, high=high, size=(4,))
        self.steps_beyond_terminated = None
        return np.array(self.state, dtype=np.float32), {}


# --------------------------------------------------------------------------- Pendulum


def angle_normalize(x):
    return ((x + np.pi) % (2 * np.pi)) - np.pi


class PendulumEnv(Env):
    """gymnasium.envs.classic_control.pendulum.PendulumEnv [UPSTREAM 1.2.1].

    Arithmetic note: the action is promoted to float64 before use (the pinned NumPy
    1.26.4 value-based promotion of ``python_float * np.float32`` scalar gives float64;
    NumPy >= 2 would keep float32).  The restatement fixes the float64 reading.
    """

    def __init__(self, render_mode=None, g=10.0):
        self.max_speed = 8
        self.max_torque = 2.0
        self.dt = 0.05
        self.g = g
        self.m = 1.0
        self.l = 1.0
        self.render_mode = render_mode
        high = np.array([1.0, 1.0, self.max_speed], dtype=np.float32)
        self.action_space = Box(low=-self.max_torque, high=self.max_torque, shape=(1,), dtype=np.float32)
        self.observation_space = Box(low=-high, high=high, dtype=np.float32)
        self.state = None
        self.last_u = None

    def step(self, u):
        th, thdot = self.state
        g, m, l, dt = self.g, self.m, self.l, self.dt
        u = float(np.clip(np.asarray(u, dtype=np.float64).reshape(-1), -self.max_torque, self.max_torque)[0])
        self.last_u = u
        # upstream `u` is still a float32 scalar here: `u**2` is a float32 scalar power (libm powf, rounded to float32) and, under the
        # NumPy-1.26 promotion the class docstring fixes, `0.001 *` of it is a float64 product.  (`th`, `thdot` are float64 scalars,
        # `l` a Python float: their `** 2` are libm pow, which is not always the correctly rounded product.)
        costs = angle_normalize(th) ** 2 + 0.1 * thdot**2 + 0.001 * float(np.float32(u) ** 2)
        newthdot = thdot + (3 * g / (2 * l) * np.sin(th) + 3.0 / (m * l**2) * u) * dt
        newthdot = np.clip(newthdot, -self.max_speed, self.max_speed)
        newth = th + newthdot * dt
        self.state = np.array([newth, newthdot])
        return self._get_obs(), -costs, False, False, {}

    def reset(self, *, seed=None, options=None):
        super().reset(seed=seed)
        high = np.array([np.pi, 1.0])
        low = -high
        self.state = self.np_random.uniform(low=low, high=high)
        self.last_u = None
        return self._get_obs(), {}

    def _get_obs(self):
        theta, thetadot = self.state
        return np.array([np.cos(theta), np.sin(theta), thetadot], dtype=np.float32)


# --------------------------------------------------------------------------- Acrobot


def wrap(x, m, M):
    diff = M - m
    while x > M:
        x = x - diff
    while x < m:
        x = x + diff
    return x


def bound(x, m, M=None):
    return min(max(x, m), M)


def rk4(derivs, y0, t):
    """gymnasium.envs.classic_control.acrobot.rk4 [UPSTREAM]; returns yout[-1][:4]."""
    Ny = len(y0)
    yout = np.zeros((len(t), Ny), np.float64)
    yout[0] = y0
    for i in np.arange(len(t) - 1):
        this = t[i]
        dt = t[i + 1] - this
        dt2 = dt / 2.0
        y0 = yout[i]
        k1 = np.asarray(derivs(y0))
        k2 = np.asarray(derivs(y0 + dt2 * k1))
        k3 = np.asarray(derivs(y0 + dt2 * k2))
        k4 = np.asarray(derivs(y0 + dt * k3))
        yout[i + 1] = y0 + dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
    return yout[-1][:4]


class AcrobotEnv(Env):
    """gymnasium.envs.classic_control.acrobot.AcrobotEnv [UPSTREAM 1.2.1], "book" dynamics."""

    dt = 0.2
    LINK_LENGTH_1 = 1.0
    LINK_LENGTH_2 = 1.0
    LINK_MASS_1 = 1.0
    LINK_MASS_2 = 1.0
    LINK_COM_POS_1 = 0.5
    LINK_COM_POS_2 = 0.5
    LINK_MOI = 1.0
    MAX_VEL_1 = 4 * np.pi
    MAX_VEL_2 = 9 * np.pi
    AVAIL_TORQUE = [-1.0, 0.0, +1]
    torque_noise_max = 0.0
    book_or_nips = "book"

    def __init__(self, render_mode=None):
        self.render_mode = render_mode
        high = np.array([1.0, 1.0, 1.0, 1.0, self.MAX_VEL_1, self.MAX_VEL_2], dtype=np.float32)
        self.observation_space = Box(low=-high, high=high, dtype=np.float32)
        self.action_space = Discrete(3)
        self.state = None

    def reset(self, *, seed=None, options=None):
        super().reset(seed=seed)
        low, high = -0.1, 0.1
        self.state = self.np_random.uniform(low=low, high=high, size=(4,)).astype(np.float32)
        return self._get_ob(), {}

    def step(self, a):
        s = self.state
        torque = self.AVAIL_TORQUE[a]
        s_augmented = np.append(s, torque)
        ns = rk4(self._dsdt, s_augmented, [0, self.dt])
        ns[0] = wrap(ns[0], -np.pi, np.pi)
        ns[1] = wrap(ns[1], -np.pi, np.pi)
        ns[2] = bound(ns[2], -self.MAX_VEL_1, self.MAX_VEL_1)
        ns[3] = bound(ns[3], -self.MAX_VEL_2, self.MAX_VEL_2)
        self.state = ns
        terminated = self._terminal()
        reward = -1.0 if not terminated else 0.0
        return self._get_ob(), reward, terminated, False, {}

    def _get_ob(self):
        s = self.state
        return np.array(
            [np.cos(s[0]), np.sin(s[0]), np.cos(s[1]), np.sin(s[1]), s[2], s[3]], dtype=np.float32
        )

    def _terminal(self):
        s = self.state
        return bool(-np.cos(s[0]) - np.cos(s[1] + s[0]) > 1.0)

    def _dsdt(self, s_augmented):
        m1 = self.LINK_MASS_1
        m2 = self.LINK_MASS_2
        l1 = self.LINK_LENGTH_1
        lc1 = self.LINK_COM_POS_1
        lc2 = self.LINK_COM_POS_2
        I1 = self.LINK_MOI
        I2 = self.LINK_MOI
        g = 9.8
        a = s_augmented[-1]
        s = s_augmented[:-1]
        theta1, theta2, dtheta1, dtheta2 = s
        cos, sin, pi = np.cos, np.sin, np.pi
        d1 = m1 * lc1**2 + m2 * (l1**2 + lc2**2 + 2 * l1 * lc2 * cos(theta2)) + I1 + I2
        d2 = m2 * (lc2**2 + l1 * lc2 * cos(theta2)) + I2
        phi2 = m2 * lc2 * g * cos(theta1 + theta2 - pi / 2.0)
        phi1 = (
            -m2 * l1 * lc2 * dtheta2**2 * sin(theta2)
            - 2 * m2 * l1 * lc2 * dtheta2 * dtheta1 * sin(theta2)
            + (m1 * lc1 + m2 * l1) * g * cos(theta1 - pi / 2)
            + phi2
        )
        if self.book_or_nips == "nips":
            ddtheta2 = (a + d2 / d1 * phi1 - phi2) / (m2 * lc2**2 + I2 - d2**2 / d1)
        else:
            ddtheta2 = (a + d2 / d1 * phi1 - m2 * l1 * lc2 * dtheta1**2 * sin(theta2) - phi2) / (
                m2 * lc2**2 + I2 - d2**2 / d1
            )
        ddtheta1 = -(d2 * ddtheta2 + phi1) / d1
        return dtheta1, dtheta2, ddtheta1, ddtheta2, 0.0


# --------------------------------------------------------------------------- MountainCar


class MountainCarEnv(Env):
    """gymnasium.envs.classic_control.mountain_car.MountainCarEnv [UPSTREAM 1.2.1]."""

    def __init__(self, render_mode=None, goal_velocity=0):
        self.min_position = -1.2
        self.max_position = 0.6
        self.max_speed = 0.07
        self.goal_position = 0.5
        self.goal_velocity = goal_velocity
        self.force = 0.001
        self.gravity = 0.0025
        self.low = np.array([self.min_position, -self.max_speed], dtype=np.float32)
        self.high = np.array([self.max_position, self.max_speed], dtype=np.float32)
        self.render_mode = render_mode
        self.action_space = Discrete(3)
        self.observation_space = Box(self.low, self.high, dtype=np.float32)
        self.state = None

    def step(self, action):
        position, velocity = self.state
        velocity += (action - 1) * self.force + math.cos(3 * position) * (-self.gravity)
        velocity = np.clip(velocity, -self.max_speed, self.max_speed)
        position += velocity
        position = np.clip(position, self.min_position, self.max_position)
        if position == self.min_position and velocity < 0:
            velocity = 0
        terminated = bool(position >= self.goal_position and velocity >= self.goal_velocity)
        reward = -1.0
        self.state = (position, velocity)
        return np.array(self.state, dtype=np.float32), reward, terminated, False, {}

    def reset(self, *, seed=None, options=None):
        super().reset(seed=seed)
        low, high = -0.6, -0.4
        self.state = np.array([self.np_random.uniform(low=low, high=high), 0])
        return np.array(self.state, dtype=np.float32), {}


class Continuous_MountainCarEnv(Env):
    """gymnasium...continuous_mountain_car.Continuous_MountainCarEnv [UPSTREAM 1.2.1].

    The state is stored back as float32 after every step upstream
    (``self.state = np.array([position, velocity], dtype=np.float32)``).
    """

    def __init__(self, render_mode=None, goal_velocity=0):
        self.min_action = -1.0
        self.max_action = 1.0
        self.min_position = -1.2
        self.max_position = 0.6
        self.max_speed = 0.07
        self.goal_position = 0.45
        self.goal_velocity = goal_velocity
        self.power = 0.0015
        self.low_state = np.array([self.min_position, -self.max_speed], dtype=np.float32)
        self.high_state = np.array([self.max_position, self.max_speed], dtype=np.float32)
        self.render_mode = render_mode
        self.action_space = Box(low=self.min_action, high=self.max_action, shape=(1,), dtype=np.float32)
        self.observation_space = Box(low=self.low_state, high=self.high_state, dtype=np.float32)
        self.state = None

    def step(self, action):
        position = float(self.state[0])
        velocity = float(self.state[1])
        a0 = float(np.asarray(action, dtype=np.float64).reshape(-1)[0])
        force = min(max(a0, self.min_action), self.max_action)
        velocity += force * self.power - 0.0025 * math.cos(3 * position)
        if velocity > self.max_speed:
            velocity = self.max_speed
        if velocity < -self.max_speed:
            velocity = -self.max_speed
        position += velocity
        if position > self.max_position:
            position = self.max_position
        if position < self.min_position:
            position = self.min_position
        if position == self.min_position and velocity < 0:
            velocity = 0
        terminated = bool(position >= self.goal_position and velocity >= self.goal_velocity)
        reward = 0
        if terminated:
            reward = 100.0
        reward -= math.pow(a0, 2) * 0.1
        self.state = np.array([position, velocity], dtype=np.float32)
        return self.state, reward, terminated, False, {}

    def reset(self, *, seed=None, options=None):
        super().reset(seed=seed)
        low, high = -0.6, -0.4
        self.state = np.array([self.np_random.uniform(low=low, high=high), 0])
        return np.array(self.state, dtype=np.float32), {}


# --------------------------------------------------------------------------- FrozenLake

MAPS = {
    "4x4": ["SFFF", "FHFH", "FFFH", "HFFG"],
    "8x8": [
        "SFFFFFFF",
        "FFFFFFFF",
        "FFFHFFFF",
        "FFFFFHFF",
        "FFFHFFFF",
        "FHHFFFHF",
        "FHFFHFHF",
        "FFFHFFFG",
    ],
}


def categorical_sample(prob_n, np_random):
    """gymnasium.envs.toy_text.utils.categorical_sample [UPSTREAM]."""
    prob_n = np.asarray(prob_n)
    csprob_n = np.cumsum(prob_n)
    return np.argmax(csprob_n > np_random.random())


class FrozenLakeEnv(Env):
    """gymnasium.envs.toy_text.frozen_lake.FrozenLakeEnv [UPSTREAM 1.2.1].

    ``P`` is built here exactly as upstream does for ``is_slippery`` True/False; the NS
    wrapper overwrites it entirely (`ns_gym/wrappers/toy_text.py:337-340`).
    """

    LEFT, DOWN, RIGHT, UP = 0, 1, 2, 3

    def __init__(self, render_mode=None, desc=None, map_name="4x4", is_slippery=True,
                 success_rate=1.0 / 3.0, reward_schedule=(1, 0, 0)):
        if desc is None:
            desc = MAPS[map_name]
        self.desc = desc = np.asarray(desc, dtype="c")
        self.nrow, self.ncol = nrow, ncol = desc.shape
        self.reward_range = (min(reward_schedule), max(reward_schedule))
        nA = 4
        nS = nrow * ncol
        self.initial_state_distrib = np.array(desc == b"S").astype("float64").ravel()
        self.initial_state_distrib /= self.initial_state_distrib.sum()
        self.P = {s: {a: [] for a in range(nA)} for s in range(nS)}
        fail_rate = (1.0 - success_rate) / 2.0

        def to_s(row, col):
            return row * ncol + col

        def inc(row, col, a):
            if a == 0:
                col = max(col - 1, 0)
            elif a == 1:
                row = min(row + 1, nrow - 1)
            elif a == 2:
                col = min(col + 1, ncol - 1)
            elif a == 3:
                row = max(row - 1, 0)
            return (row, col)

        def update_probability_matrix(row, col, action):
            new_row, new_col = inc(row, col, action)
            new_state = to_s(new_row, new_col)
            new_letter = desc[new_row, new_col]
            terminated = bytes(new_letter) in b"GH"
            reward = float(new_letter == b"G")
            return new_state, reward, terminated

        for row in range(nrow):
            for col in range(ncol):
                s = to_s(row, col)
                for a in range(4):
                    li = self.P[s][a]
                    letter = desc[row, col]
                    if letter in b"GH":
                        li.append((1.0, s, 0, True))
                    elif is_slippery:
                        for b in [(a - 1) % 4, a, (a + 1) % 4]:
                            p = success_rate if b == a else fail_rate
                            li.append((p, *update_probability_matrix(row, col, b)))
                    else:
                        li.append((1.0, *update_probability_matrix(row, col, a)))
        self.observation_space = Discrete(nS)
        self.action_space = Discrete(nA)
        self.render_mode = render_mode
        self.s = None
        self.lastaction = None

    def step(self, a):
        transitions = self.P[self.s][a]
        i = categorical_sample([t[0] for t in transitions], self.np_random)
        p, s, r, t = transitions[i]
        self.s = s
        self.lastaction = a
        return int(s), r, t, False, {"prob": p}

    def reset(self, *, seed=None, options=None):
        super().reset(seed=seed)
        self.s = categorical_sample(self.initial_state_distrib, self.np_random)
        self.lastaction = None
        return int(self.s), {"prob": 1}


# --------------------------------------------------------------------------- CliffWalking


class CliffWalkingEnv(Env):
    """gymnasium.envs.toy_text.cliffwalking.CliffWalkingEnv [UPSTREAM 1.2.1], non-slippery."""

    UP, RIGHT, DOWN, LEFT = 0, 1, 2, 3
    POSITION_MAPPING = {0: [-1, 0], 1: [0, 1], 2: [1, 0], 3: [0, -1]}

    def __init__(self, render_mode=None, is_slippery=False):
        self.shape = (4, 12)
        self.start_state_index = np.ravel_multi_index((3, 0), self.shape)
        self.nS = int(np.prod(self.shape))
        self.nA = 4
        self.is_slippery = is_slippery
        self._cliff = np.zeros(self.shape, dtype=bool)
        self._cliff[3, 1:-1] = True
        self.P = {}
        for s in range(self.nS):
            position = np.unravel_index(s, self.shape)
            self.P[s] = {a: self._calculate_transition_prob(position, a) for a in range(self.nA)}
        self.initial_state_distrib = np.zeros(self.nS)
        self.initial_state_distrib[self.start_state_index] = 1.0
        self.observation_space = Discrete(self.nS)
        self.action_space = Discrete(self.nA)
        self.render_mode = render_mode
        self.s = None
        self.lastaction = None

    def _limit_coordinates(self, coord):
        coord[0] = min(coord[0], self.shape[0] - 1)
        coord[0] = max(coord[0], 0)
        coord[1] = min(coord[1], self.shape[1] - 1)
        coord[1] = max(coord[1], 0)
        return coord

    def _calculate_transition_prob(self, current, move):
        deltas = [self.POSITION_MAPPING[move]]
        outcomes = []
        for delta in deltas:
            new_position = np.array(current) + np.array(delta)
            new_position = self._limit_coordinates(new_position).astype(int)
            new_state = np.ravel_multi_index(tuple(new_position), self.shape)
            if self._cliff[tuple(new_position)]:
                outcomes.append((1 / len(deltas), self.start_state_index, -100, False))
            else:
                terminal_state = (self.shape[0] - 1, self.shape[1] - 1)
                outcomes.append((1 / len(deltas), new_state, -1, tuple(new_position) == terminal_state))
        return outcomes

    def step(self, a):
        transitions = self.P[self.s][a]
        i = categorical_sample([t[0] for t in transitions], self.np_random)
        p, s, r, t = transitions[i]
        self.s = s
        self.lastaction = a
        return int(s), r, t, False, {"prob": p}

    def reset(self, *, seed=None, options=None):
        super().reset(seed=seed)
        self.s = categorical_sample(self.initial_state_distrib, self.np_random)
        self.lastaction = None
        return int(self.s), {"prob": 1}


# --------------------------------------------------------------------------- registry

_REGISTRY: dict[str, tuple[Any, int | None, dict]] = {
    "CartPole-v1": (CartPoleEnv, 500, {}),
    "Pendulum-v1": (PendulumEnv, 200, {}),
    "Acrobot-v1": (AcrobotEnv, 500, {}),
    "MountainCar-v0": (MountainCarEnv, 200, {}),
    "MountainCarContinuous-v0": (Continuous_MountainCarEnv, 999, {}),
    "FrozenLake-v1": (FrozenLakeEnv, 100, {"map_name": "4x4"}),
    "FrozenLake8x8-v1": (FrozenLakeEnv, 200, {"map_name": "8x8"}),
    "CliffWalking-v1": (CliffWalkingEnv, None, {}),   # registered without a TimeLimit upstream
}


def _resolve_entry_point(ep):
    if isinstance(ep, str):   # "module:attr" (the reference registers ns_gym/Bridge-v0 this way)
        import importlib

        mod, attr = ep.split(":")
        return getattr(importlib.import_module(mod), attr)
    return ep


def register(id, entry_point=None, max_episode_steps=None, **kwargs):
    if id not in _REGISTRY:
        _REGISTRY[id] = (entry_point, max_episode_steps, kwargs.get("kwargs", {}))


def make(id, max_episode_steps=None, **kwargs):
    """gymnasium.make [UPSTREAM]: TimeLimit(OrderEnforcing(PassiveEnvChecker(env))).

    OrderEnforcing / PassiveEnvChecker do no arithmetic and are not modelled.
    """
    cls, default_steps, default_kwargs = _REGISTRY[id]
    try:
        cls = _resolve_entry_point(cls)
    except Exception as e:  # an entry point whose module is absent (ns_gym/VehicleTracking-v0)
        raise KeyError(f"{id}: entry point not instantiable in the restatement") from e
    kw = dict(default_kwargs)
    kw.update(kwargs)
    env = cls(**kw)
    env.spec = EnvSpec(id, max_episode_steps or default_steps, kw)
    steps = max_episode_steps if max_episode_steps is not None else default_steps
    if steps is not None:
        env = TimeLimit(env, steps)
    return env
