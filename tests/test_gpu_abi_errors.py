"""Misuse of the C-ABI with live handles: negative code + message, no exception, no launch, and the handle stays usable."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu

ENOTBOUND, EINVAL = -77, -22   # include/nsgym_hip.h


def _handle(lib, cfg, tables, n):
    h = C.c_void_p()
    assert lib.nsg_create(C.byref(cfg), tables, len(tables), n, C.byref(h)) == 0, lib.nsg_last_error()
    return h


def test_live_handle_misuse_is_refused_and_harmless():
    import torch

    from ns_gym_amd import _abi as A
    from ns_gym_amd import _lib, make
    from ns_gym_amd.schedulers import ContinuousScheduler
    from ns_gym_amd.spec import compile_config
    from ns_gym_amd.update_functions import IncrementUpdate
    from ns_gym_amd.vec_env import VecNSEnv

    lib = _lib.load()
    tp = {"masspole": IncrementUpdate(ContinuousScheduler(), k=0.1)}
    cfg, tables, _, _ = compile_config(make("CartPole-v1"), tp)
    tables = bytes(tables)
    acts = torch.zeros(256, dtype=torch.int32, device="cuda")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    raw = _handle(lib, cfg, tables, 256)            # created, never bound
    for rc in (lib.nsg_step(raw, acts.data_ptr(), stream), lib.nsg_reset(raw, None, None, stream),
               lib.nsg_rollout(raw, acts.data_ptr(), 1, None, stream)):
        assert rc == ENOTBOUND and b"nsg_bind" in lib.nsg_last_error()
    bad = type(cfg).from_buffer_copy(cfg)           # a flag bit this library does not know; libm-exact arithmetic asked of an integer path
    bad.flags |= 0x20
    hb = C.c_void_p()
    assert lib.nsg_create(C.byref(bad), tables, len(tables), 256, C.byref(hb)) == EINVAL and b"unknown flag" in lib.nsg_last_error()
    from ns_gym_amd.update_functions import DistributionStepWiseUpdate
    fcfg, ftab, _, _ = compile_config(make("FrozenLake-v1"), {"P": DistributionStepWiseUpdate(ContinuousScheduler(), [[0.6, 0.2, 0.2]])})
    fcfg.flags |= A.F_LIBM_EXACT
    ftab = bytes(ftab)
    assert lib.nsg_create(C.byref(fcfg), ftab, len(ftab), 256, C.byref(hb)) == EINVAL and b"NSG_F_LIBM_EXACT" in lib.nsg_last_error()
    bufs = A.Buffers()                              # binding without the required rows
    assert lib.nsg_bind(raw, C.byref(bufs)) == EINVAL and b"required" in lib.nsg_last_error()
    assert lib.nsg_destroy(raw) == 0

    env = VecNSEnv(make("CartPole-v1"), tp, 256)
    env.reset(seed=0)
    h = env._h
    assert lib.nsg_step(h, None, stream) == EINVAL
    assert lib.nsg_rollout(h, acts.data_ptr(), 0, None, stream) == EINVAL
    assert lib.nsg_seed_streams(h, acts.data_ptr(), 7, stream) == EINVAL
    other = VecNSEnv(make("CartPole-v1"), tp, 256)          # not a planning copy
    assert lib.nsg_fork(h, other._h, 1, 0, stream) == EINVAL and b"SIM_ENV" in lib.nsg_last_error()
    copy = env.fork()
    assert lib.nsg_fork(h, copy._h, 1, 2, stream) == EINVAL and b"theta_mode" in lib.nsg_last_error()
    small = VecNSEnv(make("CartPole-v1"), tp, 96, is_sim_env=True)
    assert lib.nsg_fork(h, small._h, 1, 0, stream) == EINVAL and b"whole number" in lib.nsg_last_error()
    pend = VecNSEnv(make("Pendulum-v1"), {"m": IncrementUpdate(ContinuousScheduler(), k=0.01)}, 256, is_sim_env=True)
    assert lib.nsg_fork(h, pend._h, 1, 0, stream) == EINVAL and b"same configuration" in lib.nsg_last_error()
    hs = (C.c_void_p * 2)(h, None)
    ap = (C.c_void_p * 2)(acts.data_ptr(), acts.data_ptr())
    assert lib.nsg_step_group(hs, 2, ap, stream) == ENOTBOUND
    # none of the refused calls launched anything or disturbed the handle
    before = env.t.clone()
    env.step(acts)
    assert torch.equal(env.t, before + 1)
    for e in (env, other, copy, small, pend):
        e.close()


def test_read_back_copies_into_pinned_memory_and_publishes_the_sequence_number():
    """nsg_read_back: device rows -> pinned, device-mapped host memory by one small launch, completion told by the uint64 behind
    the copied bytes (what VecNSEnv.host_rows polls); misaligned / oversized requests are refused without a launch."""
    import numpy as np
    import torch

    from ns_gym_amd import _lib

    lib = _lib.load()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    src = torch.arange(4096, dtype=torch.int32, device="cuda")           # 16 KiB
    dst = torch.zeros(4096 * 4 + 16, dtype=torch.uint8).pin_memory()
    flag = dst.numpy()[4096 * 4: 4096 * 4 + 8].view(np.uint64)
    for seq in (7, 8):
        src.add_(1)
        assert lib.nsg_read_back(src.data_ptr(), dst.data_ptr(), 4096 * 4, seq, stream) == 0, lib.nsg_last_error()
        for _ in range(10_000_000):
            if int(flag[0]) == seq:
                break
        assert int(flag[0]) == seq
        np.testing.assert_array_equal(dst.numpy()[: 4096 * 4].view(np.int32), src.cpu().numpy())
    assert lib.nsg_read_back(src.data_ptr(), dst.data_ptr(), 24, 9, stream) == EINVAL            # not a multiple of 16
    assert lib.nsg_read_back(src.data_ptr() + 4, dst.data_ptr(), 32, 9, stream) == EINVAL        # misaligned source
    assert lib.nsg_read_back(src.data_ptr(), dst.data_ptr(), (1 << 20) + 16, 9, stream) == EINVAL  # beyond 1 MiB
    assert lib.nsg_read_back(None, dst.data_ptr(), 32, 9, stream) == EINVAL
    torch.cuda.synchronize()
    assert int(flag[0]) == 8                                                                       # nothing was launched
