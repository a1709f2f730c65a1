"""nsg_step / nsg_rollout only enqueue on the caller's stream (kernarg = two pointers), so a step loop
can be captured into a HIP graph and replayed - the launch-bound regime of small batches (BASELINE C2:
N = 65 536).  Replays must be indistinguishable from eager launches."""
import pytest

from tests.util import TRAJ_SPECS, make_env_from_spec

pytestmark = pytest.mark.gpu


def _vec(*a, **k):
    from ns_gym_amd.vec_env import VecNSEnv

    return VecNSEnv(*a, **k)


@pytest.mark.parametrize("specialize", [False, True])
@pytest.mark.parametrize("name", ["c2_cartpole_gravity_rw", "c3_frozenlake_step50"])
def test_captured_step_loop_replays_like_eager(name, specialize):
    import torch

    from tests.golden.make_golden import make_actions

    spec = TRAJ_SPECS[name]
    n, K, reps = 4096, 8, 6
    eager = make_env_from_spec(_vec, spec, n=n, track_returns=True, specialize=specialize)
    graph = make_env_from_spec(_vec, spec, n=n, track_returns=True, specialize=specialize)
    eager.reset(seed=11)
    graph.reset(seed=11)
    acts = torch.from_numpy(make_actions(spec["env_id"], K, n)).cuda()
    static = acts.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):   # warm-up on the capture stream (module loading must not happen inside a capture)
        graph.step(static[0])
        eager.step(acts[0])
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(K):
            graph.step(static[k])
    for r in range(reps):
        g.replay()
        for k in range(K):
            eager.step(acts[k])
    torch.cuda.synchronize()
    for row in ("t", "status", "episode", "theta", "reward", "terminated", "truncated", "rng_env", "ep_return", "last_return"):
        if graph.buf[row] is not None:
            assert torch.equal(graph.buf[row], eager.buf[row]), row
    assert torch.equal(graph.state, eager.state)
    assert graph.counters() == eager.counters()
    graph.close(); eager.close()
