"""The vectorised-NumPy restatement of BASELINE config C1 (oracle/numpy_vec.py; bench.py's `cpu_baseline.numpy_vectorised` leg,
SURVEY section 8(d) item 3) walks the same trajectories as the C oracle - and its array PCG64 is NumPy's own."""
import numpy as np

from oracle import numpy_vec as NV


def test_array_pcg64_is_numpys():
    seeds = [0, 1, 42, 2**40 + 3]
    rows = [np.zeros(len(seeds), dtype=np.uint64) for _ in range(4)]
    m = (1 << 64) - 1
    for i, s in enumerate(seeds):
        st = np.random.PCG64(np.random.SeedSequence(s)).state["state"]
        rows[0][i], rows[1][i], rows[2][i], rows[3][i] = st["state"] >> 64, st["state"] & m, st["inc"] >> 64, st["inc"] & m
    got = np.array([NV.pcg64_random(*rows) for _ in range(50)])
    want = np.array([np.random.default_rng(s).random(50) for s in seeds]).T
    np.testing.assert_array_equal(got, want)


def test_vectorised_numpy_c1_equals_the_c_oracle():
    from ns_gym_amd import make
    from ns_gym_amd.schedulers import ContinuousScheduler
    from ns_gym_amd.update_functions import IncrementUpdate
    from oracle.oracle import OracleVecEnv

    n, T = 300, 260
    o = OracleVecEnv(make("CartPole-v1"), {"masspole": IncrementUpdate(ContinuousScheduler(), k=0.1)}, n)
    v = NV.NumpyVecCartPole(n)
    o.reset(seed=5)
    v.reset(5)
    np.testing.assert_array_equal(v.obs, o.a["obs"])
    acts = np.random.default_rng(1).integers(2, size=(T, n)).astype(np.int32)
    episodes = 0
    for k in range(T):
        o.step(acts[k]); v.step(acts[k])
        np.testing.assert_allclose(v.obs, o.a["obs"], rtol=1e-6, atol=1e-6, err_msg=f"step {k}")
        np.testing.assert_array_equal(v.t, o.a["t"], err_msg=f"step {k}")
        np.testing.assert_array_equal(v.terminated, o.a["terminated"].astype(bool), err_msg=f"step {k}")
        np.testing.assert_array_equal(v.masspole, o.a["theta"][0], err_msg=f"step {k}")
        np.testing.assert_array_equal(v.env_change, o.a["env_change"][0])
        np.testing.assert_array_equal(v.reward, o.a["reward"])
        episodes += int(v.terminated.sum())
    assert episodes > 5 * n      # many resets went through the array PCG64
