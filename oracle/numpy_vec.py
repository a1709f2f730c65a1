"""TEST / BASELINE INFRASTRUCTURE ONLY - the hot path of BASELINE config C1/C5 restated as VECTORISED NumPy
(CPU baseline item 3: "vectorised NumPy ... for context").

What a NumPy user would write if the reference stepped N wrappers at once: every per-env quantity is an array over envs and one
`step(actions)` does, for all of them, what ONE `NSClassicControlWrapper.step` does for one (ns_gym/wrappers/classic_control.py:
60-100 -> ns_gym/base.py:296-363 -> gymnasium CartPoleEnv.step [UPSTREAM 1.2.1] -> TimeLimit):

    masspole' = masspole + k                      IncrementUpdate._update          single_param.py:173-175
    reject masspole' <= 0                         _constraint_checker              classic_control.py:223-228
    total_mass, polemass_length                   _dependency_resolver             classic_control.py:426-444
    Euler step, terminated, reward 1.0            CartPoleEnv.step                 [UPSTREAM] (oracle/gym_restatement.py:257-287)
    t += 1, truncated = t >= 500                  base.py:314, TimeLimit
    next call: an env whose episode ended is reset() - θ restored, t = 0, state = env.np_random.uniform(-0.05, 0.05, 4)
                                                  base.py:365-410, classic_control.py:102-109

The env streams are NumPy's own PCG64 bit for bit (state' = state * M + inc mod 2^128, XSL-RR output, `random()` =
(u64 >> 11) * 2^-53), advanced here with uint64 array arithmetic for the lanes that reset; each env's initial state comes from
`np.random.PCG64(SeedSequence(seed + i))` at reset(seed).  Checked against the C oracle in tests/test_oracle_numpy_vec.py.
Only tests/ and bench.py's cpu_baseline leg may import this."""
from __future__ import annotations

import math

import numpy as np

_M32 = np.uint64(0xFFFFFFFF)
_MULT_HI = np.uint64(2549297995355413924)     # PCG_DEFAULT_MULTIPLIER_128 (numpy/random/src/pcg64/pcg64.h [UPSTREAM])
_MULT_LO = np.uint64(4865540595714422341)
_S32, _S58, _S11 = np.uint64(32), np.uint64(58), np.uint64(11)
_U63, _U64 = np.uint64(63), np.uint64(64)


def _mul64_full(a, b):
    """(hi, lo) of the 128-bit product of two uint64 arrays, from 32-bit halves."""
    a0, a1, b0, b1 = a & _M32, a >> _S32, b & _M32, b >> _S32
    p00, p01, p10, p11 = a0 * b0, a0 * b1, a1 * b0, a1 * b1
    mid = (p00 >> _S32) + (p01 & _M32) + (p10 & _M32)
    return p11 + (p01 >> _S32) + (p10 >> _S32) + (mid >> _S32), (p00 & _M32) | ((mid & _M32) << _S32)


def pcg64_random(sh, sl, ih, il):
    """One `Generator.random()` from each stream (state hi / lo, increment hi / lo as uint64 arrays, updated in place)."""
    with np.errstate(over="ignore"):
        hi, lo = _mul64_full(sl, _MULT_LO)
        hi = hi + sh * _MULT_LO + sl * _MULT_HI
        lo2 = lo + il
        hi = hi + ih + (lo2 < lo).astype(np.uint64)
        sh[...], sl[...] = hi, lo2
        rot = hi >> _S58
        x = hi ^ lo2
        out = (x >> rot) | (x << ((_U64 - rot) & _U63))
    return (out >> _S11).astype(np.float64) * (1.0 / 9007199254740992.0)


class NumpyVecCartPole:
    """N CartPole-v1 wrappers with {"masspole": IncrementUpdate(ContinuousScheduler(), k)} stepped as arrays (next-step autoreset)."""

    def __init__(self, num_envs: int, k: float = 0.1, max_episode_steps: int = 500):
        self.N, self.k, self.max_steps = int(num_envs), float(k), int(max_episode_steps)
        n = self.N
        self.state = np.zeros((4, n))
        self.masspole = np.full(n, 0.1)
        self.t = np.zeros(n, dtype=np.int32)
        self.needs_reset = np.zeros(n, dtype=bool)
        self.rng = [np.zeros(n, dtype=np.uint64) for _ in range(4)]
        self.obs = np.zeros((n, 4), dtype=np.float32)
        self.reward = np.zeros(n, dtype=np.float32)
        self.terminated = np.zeros(n, dtype=bool)
        self.truncated = np.zeros(n, dtype=bool)
        self.env_change = np.zeros(n, dtype=np.uint8)
        self.delta_change = np.zeros(n, dtype=np.float32)

    def _draw_initial_states(self, idx):
        sub = [r[idx] for r in self.rng]
        for kk in range(4):   # uniform(-0.05, 0.05, size=4): low + (high - low) * random(), one stream draw per element
            self.state[kk, idx] = -0.05 + (0.05 - -0.05) * pcg64_random(*sub)
        self.rng[0][idx], self.rng[1][idx] = sub[0], sub[1]

    def reset(self, seed: int):
        m = (1 << 64) - 1
        for i in range(self.N):   # set-up, not the hot path: every env's own PCG64(SeedSequence(seed + i))
            st = np.random.PCG64(np.random.SeedSequence(int(seed) + i)).state["state"]
            self.rng[0][i], self.rng[1][i] = st["state"] >> 64, st["state"] & m
            self.rng[2][i], self.rng[3][i] = st["inc"] >> 64, st["inc"] & m
        self._reset_lanes(np.arange(self.N))
        return self.obs

    def _reset_lanes(self, idx):
        self._draw_initial_states(idx)
        self.masspole[idx] = 0.1
        self.t[idx] = 0
        self.obs[idx] = self.state[:, idx].T.astype(np.float32)
        self.reward[idx] = 0.0
        self.terminated[idx] = self.truncated[idx] = False
        self.env_change[idx] = 0
        self.delta_change[idx] = 0.0
        self.needs_reset[idx] = False

    def step(self, actions):
        live = ~self.needs_reset
        resetting = np.flatnonzero(self.needs_reset)
        # ---- θ (every live env fires: ContinuousScheduler) + constraint + dependency resolver
        new = self.masspole + self.k
        ok = live & ~(new <= 0)
        delta = np.where(ok, new - self.masspole, 0.0)
        masspole = np.where(ok, new, self.masspole)
        gravity, masscart, force_mag, tau, length = 9.8, 1.0, 10.0, 0.02, 0.5
        total_mass = masspole + masscart
        polemass_length = length * masspole
        # ---- CartPoleEnv.step, Euler
        x, x_dot, theta, theta_dot = self.state
        force = np.where(np.asarray(actions) == 1, force_mag, -force_mag)
        costheta, sintheta = np.cos(theta), np.sin(theta)
        temp = (force + polemass_length * np.square(theta_dot) * sintheta) / total_mass
        thetaacc = (gravity * sintheta - costheta * temp) / (length * (4.0 / 3.0 - masspole * np.square(costheta) / total_mass))
        xacc = temp - polemass_length * thetaacc * costheta / total_mass
        nx = x + tau * x_dot
        nx_dot = x_dot + tau * xacc
        ntheta = theta + tau * theta_dot
        ntheta_dot = theta_dot + tau * thetaacc
        thr = 12 * 2 * math.pi / 360
        term = (nx < -2.4) | (nx > 2.4) | (ntheta < -thr) | (ntheta > thr)
        tnew = self.t + 1
        trunc = tnew >= self.max_steps
        # ---- commit the live lanes
        self.state = np.where(live, np.stack([nx, nx_dot, ntheta, ntheta_dot]), self.state)
        self.masspole = np.where(live, masspole, self.masspole)
        self.t = np.where(live, tnew, self.t).astype(np.int32)
        self.obs[live] = self.state[:, live].T.astype(np.float32)
        self.reward = np.where(live, 1.0, 0.0).astype(np.float32)
        self.terminated = live & term
        self.truncated = live & trunc
        self.env_change = ok.astype(np.uint8)
        self.delta_change = delta.astype(np.float32)
        self.needs_reset = self.terminated | self.truncated
        # ---- the lanes whose previous step ended an episode: reset() with no seed (streams continue)
        if resetting.size:
            self._reset_lanes(resetting)
        return self.obs, self.reward, self.terminated, self.truncated


def time_c1(num_envs: int, steps: int, seed: int = 0):
    """(env-steps per second, seconds) of `steps` vectorised steps of the C1 configuration on one core."""
    import time

    env = NumpyVecCartPole(num_envs)
    env.reset(seed)
    acts = np.random.default_rng(123).integers(2, size=(8, num_envs)).astype(np.int32)
    for k in range(3):
        env.step(acts[k % 8])
    t0 = time.perf_counter()
    for k in range(steps):
        env.step(acts[k % 8])
    dt = time.perf_counter() - t0
    return num_envs * steps / dt, dt
