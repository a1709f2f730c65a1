"""Host-side robustness of the heterogeneous launch (nsg_step_group) and of the done-mask words under a masked reset.

nsg_step_group reads its members' device segments and block ranges from a per-plan table; what the library remembers about a
member list is keyed on the members' ids and valid for the members' own generations (nsg_bind / nsg_specialize of a member
re-plan, nsg_destroy of a member drops the plan - from any thread); handles that are not members never touch it.
These tests drive exactly the sequences that a cached, shared table got wrong: alternating member lists on a side stream
without synchronising, specialising a member after the group's first launch, destroying a member and creating a new env
(whose handle may reuse the address), and a destroy from another thread."""
import threading

import numpy as np
import pytest

from tests.util import TRAJ_SPECS, make_env_from_spec

pytestmark = pytest.mark.gpu


def _vec(*a, **k):
    from ns_gym_amd.vec_env import VecNSEnv

    return VecNSEnv(*a, **k)


def _acts(env, T, seed):
    import torch

    g = torch.Generator(device="cuda").manual_seed(seed)
    if env.action_is_float:
        return torch.rand((T, env.N), device="cuda", generator=g) * 4 - 2
    return torch.randint(0, env.n_actions, (T, env.N), dtype=torch.int32, device="cuda", generator=g)


def _same(a, b):
    import torch

    for row in ("theta", "t", "reward", "terminated", "truncated", "status", "episode", "obs", "cell"):
        if a.buf[row] is not None:
            assert torch.equal(a.buf[row], b.buf[row]), row


def _pair(name, n, **kw):
    a, b = (make_env_from_spec(_vec, TRAJ_SPECS[name], n=n, **kw) for _ in range(2))
    a.reset(seed=3); b.reset(seed=3)
    return a, b


def test_alternating_groups_on_a_side_stream_without_sync():
    import torch

    from ns_gym_amd.vec_env import step_group

    T = 40
    (p, p_ref), (a, a_ref) = _pair("c4_pendulum_m_inc", 40000, specialize=False), _pair("c4_acrobot_mass2_inc", 30000, specialize=False)
    (c, c_ref), (f, f_ref) = _pair("c1_cartpole_masspole_inc", 50000, specialize=False), _pair("c3_frozenlake_step50", 20000, specialize=False)
    acts = {e: _acts(e, T, k) for k, e in enumerate((p, a, c, f))}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for k in range(T):   # two member lists alternate back to back: no launch may see the other's segments
            step_group([p, a], [acts[p][k], acts[a][k]])
            step_group([f, c, a], [acts[f][k], acts[c][k], acts[a][k]]) if k % 2 else step_group([c, f], [acts[c][k], acts[f][k]])
    side.synchronize()
    for k in range(T):
        p_ref.step(acts[p][k]); a_ref.step(acts[a][k]); c_ref.step(acts[c][k]); f_ref.step(acts[f][k])
        if k % 2:
            a_ref.step(acts[a][k])
    torch.cuda.synchronize()
    for x, y in ((p, p_ref), (a, a_ref), (c, c_ref), (f, f_ref)):
        _same(x, y)
    for e in (p, a, c, f, p_ref, a_ref, c_ref, f_ref):
        e.close()


def test_group_follows_specialise_destroy_and_recreate():
    import torch

    from ns_gym_amd.vec_env import step_group

    T = 12
    (p, p_ref), (c, c_ref) = _pair("c4_pendulum_m_inc", 9000, specialize=False), _pair("c1_cartpole_masspole_inc", 7000, specialize=False)
    ap, ac = _acts(p, 3 * T, 1), _acts(c, 3 * T, 2)
    for k in range(T):
        step_group([p, c], [ap[k], ac[k]])
    p.specialize(); c.specialize()           # after the group's first launches: the group must pick the new units up
    for k in range(T, 2 * T):
        step_group([p, c], [ap[k], ac[k]])
    for k in range(2 * T):
        p_ref.step(ap[k]); c_ref.step(ac[k])
    _same(p, p_ref); _same(c, c_ref)
    # destroy a member from ANOTHER thread, create a new env (its handle may land on the freed address): the remembered
    # plan for [p, c] must not be applied to [p, c2]
    th = threading.Thread(target=c.close)
    th.start(); th.join()
    c2 = make_env_from_spec(_vec, TRAJ_SPECS["c3_frozenlake_step50"], n=5000, specialize=False)
    c2_ref = make_env_from_spec(_vec, TRAJ_SPECS["c3_frozenlake_step50"], n=5000, specialize=False)
    c2.reset(seed=9); c2_ref.reset(seed=9)
    a2 = _acts(c2, T, 5)
    for k in range(T):
        step_group([p, c2], [ap[2 * T + k], a2[k]])
        p_ref.step(ap[2 * T + k]); c2_ref.step(a2[k])
    torch.cuda.synchronize()
    _same(p, p_ref); _same(c2, c2_ref)
    for e in (p, p_ref, c_ref, c2, c2_ref):
        e.close()


def test_plan_survives_unrelated_handles_and_many_member_lists():
    """What round 2's advisor found: every nsg_bind / nsg_destroy of ANY handle invalidated every plan (a planning copy made
    per simulation cost the next group launch a device synchronisation + a blocking copy), and a slot's table was overwritten
    after four re-plans.  Now: a plan is untouched by handles that are not its members, and more member lists than there are
    plan slots (16) alternate with every launch reading its own members' segments."""
    import torch

    from ns_gym_amd.vec_env import step_group, step_group_kind

    T = 6
    (p, p_ref), (a, a_ref) = _pair("c4_pendulum_m_inc", 3000, specialize=False), _pair("c4_acrobot_mass2_inc", 2000, specialize=False)
    ap, aa = _acts(p, 40 * T, 1), _acts(a, 40 * T, 2)
    assert step_group_kind([p, a]) == "unplanned"
    step_group([p, a], [ap[0], aa[0]])
    assert step_group_kind([p, a]) == "generic"
    k = 1
    for _ in range(5):   # unrelated handles come and go: bind + destroy, a planning copy, a specialised batch
        other = make_env_from_spec(_vec, TRAJ_SPECS["c1_cartpole_masspole_inc"], n=512, specialize=False)
        other.reset(seed=1)
        copy = other.fork()
        assert step_group_kind([p, a]) == "generic"          # still planned
        step_group([p, a], [ap[k], aa[k]]); k += 1
        copy.close(); other.close()
        assert step_group_kind([p, a]) == "generic"
    # 20 distinct member lists (more than the 16 plan slots) round-robin, [p, a] in between
    smalls = [make_env_from_spec(_vec, TRAJ_SPECS["c3_frozenlake_step50"], n=300 + 10 * j, specialize=False) for j in range(20)]
    refs = [make_env_from_spec(_vec, TRAJ_SPECS["c3_frozenlake_step50"], n=300 + 10 * j, specialize=False) for j in range(20)]
    for e in smalls + refs:
        e.reset(seed=4)
    sa = [_acts(e, 3, 50 + j) for j, e in enumerate(smalls)]
    for r in range(3):
        for j, e in enumerate(smalls):
            step_group([e, p], [sa[j][r], ap[k]])
            step_group([p, a], [ap[k + 1], aa[k]])
            refs[j].step(sa[j][r])
            k += 2
    for q in range(k):
        p_ref.step(ap[q])
    n_a = 1 + 5 + 60
    for q in range(n_a):
        a_ref.step(aa[[0, 1, 2, 3, 4, 5][q] if q < 6 else 6 + 2 * (q - 6)])
    torch.cuda.synchronize()
    _same(p, p_ref); _same(a, a_ref)
    for e, r in zip(smalls, refs):
        _same(e, r)
    for e in [p, p_ref, a, a_ref] + smalls + refs:
        e.close()


def test_group_launch_inside_a_graph_capture():
    """A planned member list is graph-capturable (the launch reads kernel arguments and an immutable table); a list that would
    have to be planned during the capture is refused with an error instead of synchronising inside it."""
    import torch

    from ns_gym_amd._lib import NsgError
    from ns_gym_amd.vec_env import step_group

    (p, p_ref), (c, c_ref) = _pair("c4_pendulum_m_inc", 5000, specialize=False), _pair("c1_cartpole_masspole_inc", 4000, specialize=False)
    ap, ac = _acts(p, 1, 1)[0], _acts(c, 1, 2)[0]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = ap + 0                       # the capture holds at least one node
        with pytest.raises(NsgError, match="capturing"):
            step_group([p, c], [ap, ac])
    step_group([p, c], [ap, ac])           # plans (outside any capture)
    p_ref.step(ap); c_ref.step(ac)
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        step_group([p, c], [ap, ac])
        step_group([p, c], [ap, ac])
    for _ in range(5):
        g2.replay()
        for _ in range(2):
            p_ref.step(ap); c_ref.step(ac)
    del keep
    torch.cuda.synchronize()
    _same(p, p_ref); _same(c, c_ref)
    for e in (p, p_ref, c, c_ref):
        e.close()


def test_group_refuses_duplicates():
    from ns_gym_amd._lib import NsgError
    from ns_gym_amd.vec_env import step_group

    p = make_env_from_spec(_vec, TRAJ_SPECS["c4_pendulum_m_inc"], n=1000, specialize=False)
    p.reset(seed=0)
    a = _acts(p, 1, 0)
    with pytest.raises(NsgError, match="listed twice"):
        step_group([p, p], [a[0], a[0]])
    p.close()


@pytest.mark.parametrize("name", ["c1_cartpole_masspole_inc", "c3_frozenlake_step50"])
def test_masked_reset_keeps_the_other_envs_done_bits(name):
    """nsg_reset(mask): the ballot words that nsg_compact_done expands must lose exactly the bits of the envs that were reset."""
    import torch

    n = 10007
    env = make_env_from_spec(_vec, TRAJ_SPECS[name], n=n, specialize=False)
    env.reset(seed=1)
    acts = _acts(env, 40, 3)
    for k in range(40):
        env.step(acts[k])
    done = (env.terminated | env.truncated).cpu().numpy()
    assert done.sum() > 50
    rng = np.random.default_rng(0)
    mask = rng.random(n) < 0.5                      # resets done and not-done envs alike, lane 0 of a wavefront or not
    env.reset(mask=torch.from_numpy(mask).cuda())
    want = np.flatnonzero(done & ~mask)
    got = np.sort(env.done_indices().cpu().numpy())
    np.testing.assert_array_equal(got, want)
    env.close()
