"""The prebuilt units on the GPU: the BASELINE configurations' specialised kernels come from ns_gym_amd/prebuilt/ (built and
inspected with the library, `__graft_entry__.build()`), not from the box's runtime compiler - and they are the same kernels:
prebuilt unit == hiprtc unit == generic kernels, bit for bit."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name,n", [("c1", 1 << 20), ("c2", 1 << 16), ("c3", 1 << 18), ("pend", 1 << 18)])
def test_prebuilt_equals_hiprtc_equals_generic(name, n, monkeypatch):
    from ns_gym_amd import workloads as W

    a = W.build(name, n, specialize=True, seed=5)
    assert a.kernels == "config-specialised (prebuilt)", a.kernels
    g = W.build(name, n, specialize=False, seed=5)
    assert g.kernels == "generic (precompiled)"
    # the same sources through the runtime compiler: an extra option that defines nothing the kernels read changes the unit's key
    # (NSG_SPEC_FLAGS is part of it), so neither the prebuilt directory nor the process cache has it
    monkeypatch.setenv("NSG_SPEC_FLAGS", "-DNSG_KEY_SALT_FOR_THIS_TEST=1")
    monkeypatch.setenv("NSG_SPEC_CACHE", "off")
    h = W.build(name, n, specialize=True, seed=5)
    assert h.kernels == "config-specialised (hiprtc)", h.kernels
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    K = 24
    acts = [W.random_actions(a, generator=gen) for _ in range(K)]
    for k in range(K):
        a.step(acts[k]); g.step(acts[k]); h.step(acts[k])
    head = a._arena_head
    assert torch.equal(a._arena[:head], h._arena[:head]), "prebuilt unit differs from the hiprtc unit"
    assert torch.equal(a._arena[:head], g._arena[:head]), "prebuilt unit differs from the generic kernels"
    ra, rg = a.rollout(torch.stack(acts)), g.rollout(torch.stack(acts))
    for key in ra:
        assert torch.equal(ra[key], rg[key]), key
    assert torch.equal(a._arena[:head], g._arena[:head])
    for e in (a, g, h):
        e.close()


def test_without_the_runtime_compiler_the_baseline_configs_still_run_specialised():
    """NSG_NO_HIPRTC=1 (a box without libhiprtc): C1, C2, C3 and the C4 pair come up specialised - from the prebuilt directory -
    while a configuration nobody prebuilt stays generic (and says so)."""
    code = r'''
import warnings
import torch
from ns_gym_amd import make, workloads as W
from ns_gym_amd.schedulers import PeriodicScheduler
from ns_gym_amd.update_functions import IncrementUpdate
from ns_gym_amd.vec_env import VecNSEnv, step_group, step_group_kind, rollout_group
for name in ("c1", "c2", "c3"):
    e = W.build(name, specialize=True)
    assert e.kernels == "config-specialised (prebuilt)", (name, e.kernels)
    e.step(W.random_actions(e)); e.close()
big = W.build("c1", 1 << 24, specialize=True)      # the roofline_hbm_resident unit (state rows streamed)
assert big.kernels == "config-specialised (prebuilt)"; big.close()
p, a = W.build("pend", specialize=True), W.build("acro", specialize=True)
step_group([p, a], [W.random_actions(p), W.random_actions(a)])
assert step_group_kind([p, a]) == "specialised (prebuilt)", step_group_kind([p, a])
rollout_group([p, a], [torch.stack([W.random_actions(e) for _ in range(4)]) for e in (p, a)])
# the fused policy rollouts of the BASELINE configurations come from the prebuilt directory as well
from ns_gym_amd.policies import EpisodeAccounts, UniformRandom
for name in ("c1", "c2", "c3", "pend", "acro"):
    e = W.build(name, specialize=True)
    e.rollout_policy(UniformRandom(seed=1), 8, accounts=EpisodeAccounts(e, gamma=0.9))
    assert e.policy_kernels == "config-specialised", (name, e.policy_kernels)
    e.close()
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    other = VecNSEnv(make("CartPole-v1"), {"force_mag": IncrementUpdate(PeriodicScheduler(7), k=0.3)}, 1 << 16)
assert other.kernels == "generic (precompiled)" and any("not available" in str(x.message) for x in w)
torch.cuda.synchronize()
print("ok")
'''
    env = dict(os.environ, NSG_NO_HIPRTC="1", NSG_SPEC_CACHE="off")
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), p.stderr[-3000:]


def test_bench_line_names_the_prebuilt_unit_without_the_runtime_compiler():
    import json

    env = dict(os.environ, NSG_NO_HIPRTC="1", NSG_SPEC_CACHE="off")
    p = subprocess.run([sys.executable, "bench.py", "--steps", "40", "--warmup", "10", "--no-cpu-baseline", "--no-all-configs"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert d["config"]["kernels"] == "config-specialised (prebuilt)" and d["config"]["envs_per_gpu"] == 1 << 20
    assert d["roofline_hbm_resident"]["kernels"] == "config-specialised (prebuilt)"
    assert d["roofline"]["frac"] > 0.5
