"""Grid envs: bits 1-7 of the status byte name which of the config's distributions the wrapper's P table holds
(NSG_ST_TABLE_*, include/nsgym_hip.h), so that a step can skip the float64 `table_prob` rows.  The rows stay authoritative:
these tests hold the bits against the rows after construction, fires, resets and forks, and show that a byte which makes no
statement (or names nothing valid for the config) gives the same trajectory from the rows."""
import pytest

pytestmark = pytest.mark.gpu

ROWS = ("cell", "t", "theta", "table_prob", "rng_env", "prob", "reward", "terminated", "truncated", "env_change", "delta_change",
        "last_return", "last_length", "cursor")
INITIAL, LIST0 = 1, 2


def _c3(n, **kw):
    from ns_gym_amd import workloads

    return workloads.build("c3", n, **kw)


def _acts(env, T, seed=0):
    import torch

    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randint(0, env.n_actions, (T, env.N), dtype=torch.int32, device="cuda", generator=g)


def _hint(env):
    return env.buf["status"] >> 1


def _check_bits_against_rows(env, initial, entries):
    """Every env whose byte names a distribution holds exactly that distribution in its table_prob rows."""
    import torch

    h = _hint(env)
    tp = env.table_prob
    ini = torch.tensor(initial, dtype=torch.float64, device="cuda")
    assert torch.equal(tp[:, h == INITIAL], ini[:, None].expand(-1, int((h == INITIAL).sum())))
    for j, e in enumerate(entries):
        m = h == LIST0 + j
        assert torch.equal(tp[:, m], torch.tensor(e, dtype=torch.float64, device="cuda")[:, None].expand(-1, int(m.sum())))
    assert int(((h != INITIAL) & (h >= LIST0 + len(entries))).sum()) == 0
    return h


@pytest.mark.parametrize("specialize", [False, True])
def test_bits_follow_the_table_through_fires_and_resets(specialize):
    import torch

    env = _c3(3000, specialize=specialize)
    assert env.specialized == specialize
    assert torch.all(_hint(env) == INITIAL) and torch.all(env.buf["status"] & 1 == 0)
    acts = _acts(env, 130)
    seen_list = False
    for k in range(130):                                   # episodes truncate at 100: fires at t = 50, autoresets, second fires
        env.step(acts[k])
        if k in (10, 49, 50, 51, 75, 101, 129):
            h = _check_bits_against_rows(env, [1.0, 0.0, 0.0], [[0.6, 0.2, 0.2]])
            seen_list |= bool((h == LIST0).any())
    assert seen_list and bool((_hint(env) == INITIAL).any())   # both kinds are in the batch (envs that never reached t = 50)
    env.reset(seed=5)                                       # an explicit reset keeps the table and therefore the bits
    h = _check_bits_against_rows(env, [1.0, 0.0, 0.0], [[0.6, 0.2, 0.2]])
    assert bool((h == LIST0).any()) and torch.all(env.buf["status"] & 1 == 0)


@pytest.mark.parametrize("specialize", [False, True])
@pytest.mark.parametrize("byte", [0, 100])
def test_a_byte_that_names_nothing_steps_from_the_rows(specialize, byte):
    """Same envs, same actions: one batch with its bits, one whose bits were cleared (0) or set to an entry the one-entry
    list does not have (100) half way - every row but the status byte stays identical, and the bits come back with the next fire."""
    import torch

    a, b = _c3(2000, specialize=specialize), _c3(2000, specialize=specialize)
    acts = _acts(a, 160, seed=3)
    for k in range(60):
        a.step(acts[k])
        b.step(acts[k])
    b.buf["status"].copy_((b.buf["status"] & 1) | (byte << 1))
    for k in range(60, 160):
        a.step(acts[k])
        b.step(acts[k])
        if k in (60, 61, 100, 159):
            for row in ROWS:
                if a.buf[row] is not None:
                    assert torch.equal(a.buf[row], b.buf[row]), (row, k)
            assert torch.equal(a.buf["status"] & 1, b.buf["status"] & 1)
    m = _hint(b) == LIST0
    assert bool(m.any()) and torch.equal(_hint(a)[m], _hint(b)[m])


def test_a_caller_that_writes_the_table_clears_the_bits():
    """The documented contract for an outside writer of table_prob: clear the bits, and the step samples from the written rows."""
    import torch

    env = _c3(1024)
    env._table_prob_blocked[:, 0, :] = 0.25
    env._table_prob_blocked[:, 1, :] = 0.5
    env._table_prob_blocked[:, 2, :] = 0.25
    env.buf["status"].copy_(env.buf["status"] & 1)
    acts = _acts(env, 5, seed=9)
    for k in range(5):
        env.step(acts[k])
        # (five steps from the start cell: no env has needed a reset yet, so every env has sampled a move)
        assert bool(torch.isin(env.prob, torch.tensor([0.25, 0.5], device="cuda")).all()) and bool((env.prob == 0.5).any())
    assert torch.all(_hint(env) == 0)


def test_computed_distributions_make_no_statement():
    """DistributionDecrementUpdate computes its values: after an env's first fire its byte says 'read the rows'."""
    import torch

    from ns_gym_amd import make
    from ns_gym_amd.schedulers import PeriodicScheduler
    from ns_gym_amd.update_functions import DistributionDecrementUpdate
    from ns_gym_amd.vec_env import VecNSEnv

    env = VecNSEnv(make("FrozenLake-v1", map_name="8x8"), {"P": DistributionDecrementUpdate(PeriodicScheduler(period=4), 0.05)}, 512,
                   change_notification=True, delta_change_notification=True)
    env.reset(seed=0)
    assert torch.all(_hint(env) == INITIAL)
    acts = _acts(env, 6, seed=1)
    for k in range(6):
        env.step(acts[k])
    fired_once = env.table_prob[0] != 1.0
    assert bool(fired_once.any()) and torch.all(_hint(env)[fired_once] == 0)


@pytest.mark.parametrize("env_id,kw,theta_mode", [("FrozenLake-v1", {"map_name": "8x8"}, 0), ("FrozenLake-v1", {"map_name": "8x8"}, 1),
                                                   ("CliffWalking-v1", {}, 0), ("CliffWalking-v1", {}, 1)])
def test_planning_copies_carry_the_bits_of_the_table_they_step_with(env_id, kw, theta_mode):
    import torch

    from ns_gym_amd import make
    from ns_gym_amd.schedulers import DiscreteScheduler
    from ns_gym_amd.update_functions import DistributionStepWiseUpdate
    from ns_gym_amd.vec_env import VecNSEnv

    nd = 4 if env_id.startswith("Cliff") else 3
    ini = [1.0] + [0.0] * (nd - 1)
    entries = [[0.7] + [0.3 / (nd - 1)] * (nd - 1), [0.4] + [0.6 / (nd - 1)] * (nd - 1)]
    env = VecNSEnv(make(env_id, **kw), {"P": DistributionStepWiseUpdate(DiscreteScheduler({3, 9}), entries)}, 1500,
                   change_notification=True, delta_change_notification=True, initial_prob_dist=ini)
    env.reset(seed=2)
    acts = _acts(env, 40, seed=4)
    for k in range(12):
        env.step(acts[k])
    _check_bits_against_rows(env, ini, entries)
    copy = env.fork(theta_mode=theta_mode, entropy=11)
    _check_bits_against_rows(copy, ini, entries)
    for k in range(12, 40):
        copy.step(acts[k])
    _check_bits_against_rows(copy, ini, entries)
