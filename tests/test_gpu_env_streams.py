"""The classic-control envs' np_random streams are never stored: the kernels re-derive env i's stream at episode e from its
seed by PCG64 jump-ahead (nsg_rng.hip.h).  Every way of (re-)seeding must still give, reset for reset, the initial states the
reference's per-env `np.random.Generator` gives - checked against the oracle, which keeps a plain PCG64 record per env:

  * reset(seed=int) / an affine seed array (the affine form: nothing stored per env) over many episodes per env,
  * an arbitrary seed array (per-env (seed, key) records), a masked re-seed of a subset, reset() without a seed (streams
    continue), seed_streams(),
  * the jump itself at large draw counts, against NumPy's own `advance`."""
import numpy as np
import pytest

from tests.util import TRAJ_SPECS, GpuView, OracleView, compare_views, make_env_from_spec

pytestmark = pytest.mark.gpu


def _vec(*a, **k):
    from ns_gym_amd.vec_env import VecNSEnv

    return VecNSEnv(*a, **k)


def _pair(name, n, **kw):
    from oracle.oracle import OracleVecEnv

    spec = TRAJ_SPECS[name]
    return GpuView(make_env_from_spec(_vec, spec, n=n, **kw)), OracleView(make_env_from_spec(OracleVecEnv, spec, n=n))


def _run(g, o, acts, k0, k1, tag):
    for k in range(k0, k1):
        compare_views(g.step(acts[k]), o.step(acts[k]), False, f"{tag}: step {k}")


@pytest.mark.parametrize("name", ["c1_cartpole_masspole_inc", "c4_pendulum_m_inc", "mountaincar", "c4_acrobot_mass2_inc"])
@pytest.mark.parametrize("specialize", [False, True])
def test_every_way_of_seeding_gives_the_reference_streams(name, specialize):
    import torch

    n, T = 3000, 60 if "acrobot" in name else 120
    g, o = _pair(name, n, specialize=specialize)
    rng = np.random.default_rng(7)
    if g.env.action_is_float:
        acts = (rng.random((6 * T, n)) * 4 - 2).astype(np.float32)
    else:
        acts = rng.integers(0, g.env.n_actions, size=(6 * T, n)).astype(np.int32)
    # 1. affine: reset(seed=int)
    g.env.reset(seed=1234); o.env.reset(seed=1234)
    assert int(g.env.buf["rng_env"][0].item()) < 0            # descriptor word 0 has the AFFINE bit (bit 63) set
    _run(g, o, acts, 0, T, "reset(seed=int)")
    # 2. an arbitrary seed array: per-env records
    seeds = rng.integers(0, 2**62, size=n).astype(np.uint64)
    compare_views(g.reset(seeds), o.reset(seeds), False, "reset(seed=array)")
    assert int(g.env.buf["rng_env"][0].item()) >= 0
    _run(g, o, acts, T, 2 * T, "reset(seed=array)")
    # 3. masked re-seed of a subset (the others keep their streams AND their running episodes)
    mask = rng.random(n) < 0.3
    seeds2 = rng.integers(0, 2**62, size=n).astype(np.uint64)
    g.env.reset(seed=seeds2, mask=torch.from_numpy(mask)); o.env.reset(seed=seeds2, mask=mask)
    _run(g, o, acts, 2 * T, 3 * T, "masked re-seed")
    # 4. reset() without a seed: every stream continues with its next episode
    g.env.reset(); o.env.reset()
    compare_views(g._out(), o._out(), False, "reset()")
    _run(g, o, acts, 3 * T, 4 * T, "reset()")
    # 5. back to the affine form, then a masked affine re-seed (leaves it) and seed_streams
    g.env.reset(seed=np.arange(n, dtype=np.uint64) + np.uint64(99)); o.env.reset(seed=99)
    assert int(g.env.buf["rng_env"][0].item()) < 0
    _run(g, o, acts, 4 * T, 4 * T + T // 2, "affine array")
    g.env.reset(seed=5, mask=torch.from_numpy(mask)); o.env.reset(seed=5, mask=mask)
    _run(g, o, acts, 4 * T + T // 2, 5 * T, "masked affine re-seed")
    s3 = rng.integers(0, 2**62, size=n).astype(np.uint64)
    g.env.seed_streams(s3, which="env"); o.env.seed_streams(s3, which=0)
    _run(g, o, acts, 5 * T, 6 * T, "seed_streams")
    g.env.close()


def test_many_episodes_per_env_stay_on_the_stream():
    """CartPole with a fast-growing masspole: ~10-step episodes, so 3000 steps walk every env through ~300 resets - two digits of the
    jump table and their carries - and the planning copy's own stream (entropy + i, key 7001) through as many."""
    import torch

    n, T = 2048, 3000
    g, o = _pair("c1_cartpole_masspole_inc", n)
    g.env.reset(seed=31); o.env.reset(seed=31)
    acts = np.random.default_rng(3).integers(0, 2, size=(T, n)).astype(np.int32)
    for k in range(T):
        g.env.step(torch.from_numpy(acts[k]))
        o.env.step(acts[k])
        if k % 500 == 499:
            compare_views(g._out(), o._out(), False, f"step {k}")
    ep = (g.env.buf["episode"] >> 1).cpu().numpy()
    assert ep.min() > 40 and ep.max() > 100           # (counts beyond one table digit: the large-count test below)
    g.env.close()


def test_jump_ahead_at_large_draw_counts_against_numpy():
    """Episode counts far beyond what a test can step through: set the episode word by hand, reset() without a seed, and compare
    the drawn initial state with NumPy's generator advanced to the same draw."""
    import torch

    from ns_gym_amd import make
    from ns_gym_amd.vec_env import VecNSEnv

    n = 512
    env = VecNSEnv(make("CartPole-v1"), {}, n)
    env.reset(seed=1000)
    counts = np.concatenate([[0, 1, 63, 64, 65535, 65536, 2**24 - 1, 2**24, 2**30 + 12345, 2**31 - 2],
                             np.random.default_rng(0).integers(0, 2**31 - 1, size=n - 10)]).astype(np.int64)
    env.buf["episode"].copy_(torch.from_numpy((counts << 1).astype(np.int32)).cuda())
    env.reset()                                        # reset(seed=None): episode `count` of each env's stream
    got = env.phys.cpu().numpy().T                     # [n, 4] float64 initial states
    for i in list(range(12)) + [100, 511]:
        bg = np.random.PCG64(np.random.SeedSequence(1000 + i))
        if counts[i]:
            bg.advance(4 * int(counts[i]))
        want = np.random.Generator(bg).uniform(-0.05, 0.05, size=4)
        np.testing.assert_array_equal(got[i], want, err_msg=f"env {i}, episode {counts[i]}")
    np.testing.assert_array_equal((env.buf["episode"].cpu().numpy().view(np.uint32) >> 1).astype(np.int64), counts + 1)
    env.close()
