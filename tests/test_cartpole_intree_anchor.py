"""SURVEY §8 row a20, the part the reference tree CAN pin: CartPole's Euler step against the one integrator formula
the reference itself holds - `NSCartPoleV0.transition` (ns_gym/benchmark_algorithms/rats-experiments/code/envs/
nscartpole_v0.py:84-115), driven by tests/golden/make_cartpole_anchor.py at time 0 / actions {0, 4} /
is_model_dynamic=False, where it reduces to CartPole-v1's step with the same constants (:24-36).

12 theta settings x 512 states.  Three implementations are held against those numbers: the Python restatement of
gymnasium's CartPoleEnv (oracle/gym_restatement.py), the C oracle (through the full wrapper step: theta installed by
IncrementUpdate fns firing at t = 0) and - `-m gpu` - the HIP kernels through the C-ABI (`nsg_step`).

Allowance, and why it is not bit for bit: the legacy file multiplies left to right
(`polemass_length * theta_dot * theta_dot * sintheta`, `masspole * costheta * costheta / total_mass`, :99-100) where
gymnasium 1.2.1 - which the restatement, the oracle and the kernels follow - squares first (`np.square(...)`), and it
calls `math.sin / math.cos` where they call NumPy's / libm's / the kernels' own sincos (each within 1 ulp).  The
re-association moves `temp` and `thetaacc` by a few ulps; `x` and `theta` (x + tau * x_dot, theta + tau * theta_dot)
have no such term and must be BIT-IDENTICAL.  Asserted: x, theta exact; x_dot, theta_dot within 8 * 2^-52 * scale of
the anchor's value, scale = max(|v|, |v'|, |v' - v|) (the velocity update v' = v + tau * acc cancels, so the unit is the
size of its operands, not of its result); measured here: 98.3 % of the 12 288 velocities bit-identical, worst 2.5 units;
`done` identical (no sampled state sits within an ulp of a threshold).  PENDULUM / ACROBOT / MOUNTAINCAR
have no in-tree formula: their integrators stay parity-unpinned.
"""
import numpy as np
import pytest

from tests.util import load

NAMES = ("gravity", "masscart", "masspole", "force_mag", "tau", "length")
UNITS = 8          # x 2^-52 x max(|v|, |v'|, |v' - v|)


def _anchor():
    return load("cartpole_intree_anchor.npz")


def _check(next_state, done, rec, s, tag):
    want = rec["next_state"][s]
    # position and angle: one multiply-add of values the anchor and the restatement share -> identical bits
    np.testing.assert_array_equal(next_state[:, 0], want[:, 0], err_msg=f"{tag}: x")
    np.testing.assert_array_equal(next_state[:, 2], want[:, 2], err_msg=f"{tag}: theta")
    for col, name in ((1, "x_dot"), (3, "theta_dot")):
        v0 = rec["states"][:, col]
        scale = np.maximum(np.maximum(np.abs(v0), np.abs(want[:, col])), np.abs(want[:, col] - v0))
        err = np.abs(next_state[:, col] - want[:, col]) / (2.0 ** -52 * scale)
        assert err.max() <= UNITS, f"{tag}: {name} differs by {err.max():.1f} units (env {int(err.argmax())})"
    np.testing.assert_array_equal(done.astype(np.uint8), rec["done"][s], err_msg=f"{tag}: done")


def test_fixture_is_what_the_generator_describes():
    rec = _anchor()
    assert rec["states"].shape == (512, 4) and rec["thetas"].shape == (12, 6) and rec["next_state"].shape == (12, 512, 4)
    # x' = x + tau * x_dot, theta' = theta + tau * theta_dot: reproducible from the inputs alone, in any implementation
    for s, th in enumerate(rec["thetas"]):
        np.testing.assert_array_equal(rec["next_state"][s][:, 0], rec["states"][:, 0] + th[4] * rec["states"][:, 1])
        np.testing.assert_array_equal(rec["next_state"][s][:, 2], rec["states"][:, 2] + th[4] * rec["states"][:, 3])
    assert 0 < rec["done"].sum() < rec["done"].size   # both outcomes are covered


def test_python_restatement_matches_the_intree_formula():
    from oracle import gym_restatement as G

    rec = _anchor()
    for s, th in enumerate(rec["thetas"]):
        env = G.CartPoleEnv()
        env.gravity, env.masscart, env.masspole, env.force_mag, env.tau, env.length = (float(v) for v in th)
        env.total_mass = env.masspole + env.masscart          # _dependency_resolver, classic_control.py:426-444
        env.polemass_length = env.length * env.masspole
        nxt = np.zeros((512, 4))
        done = np.zeros(512, dtype=bool)
        for i in range(512):
            env.state = rec["states"][i].copy()
            env.steps_beyond_terminated = None
            _, _, term, _, _ = env.step(int(rec["actions"][i]))
            nxt[i], done[i] = env.state, term
        _check(nxt, done, rec, s, f"gym_restatement, theta setting {s}")


def _params(deltas):
    from ns_gym_amd.schedulers import ContinuousScheduler
    from ns_gym_amd.update_functions import IncrementUpdate

    # every theta is installed by the wrapper path itself: IncrementUpdate(+delta) firing at t = 0 only
    return {n: IncrementUpdate(ContinuousScheduler(start=0, end=0), k=float(d)) for n, d in zip(NAMES, deltas)}


def test_c_oracle_matches_the_intree_formula():
    from ns_gym_amd import make
    from oracle.oracle import OracleVecEnv

    rec = _anchor()
    for s, deltas in enumerate(rec["deltas"]):
        env = OracleVecEnv(make("CartPole-v1"), _params(deltas), 512)
        env.reset(seed=0)
        env.a["phys"][:4, :] = rec["states"].T
        env.step(rec["actions"])
        np.testing.assert_array_equal(env.a["theta"][:6, 0], rec["thetas"][s])   # theta' = default + delta, as the generator computed it
        _check(env.a["phys"][:4].T.copy(), env.a["terminated"], rec, s, f"C oracle, theta setting {s}")


@pytest.mark.gpu
@pytest.mark.parametrize("specialize", [False, True])
def test_hip_step_matches_the_intree_formula(specialize):
    import torch

    from ns_gym_amd import make
    from ns_gym_amd.vec_env import VecNSEnv

    rec = _anchor()
    for s, deltas in enumerate(rec["deltas"]):
        env = VecNSEnv(make("CartPole-v1"), _params(deltas), 512, specialize=specialize)
        env.reset(seed=0)
        env.set_phys(torch.from_numpy(rec["states"].T.copy()))
        _, _, term, _, _ = env.step(torch.from_numpy(rec["actions"]).cuda())
        np.testing.assert_array_equal(env.theta[:, 0].cpu().numpy(), rec["thetas"][s])
        _check(env.phys.cpu().numpy().T.copy(), term.cpu().numpy(), rec, s, f"HIP nsg_step (specialize={specialize}), theta setting {s}")
        env.close()
