"""N > 1 path on CPU: world_size-2 gloo run of the sharding helpers and of the one collective of
the path (all-gather of per-env episode returns)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_partition_everything():
    from ns_gym_amd.distributed import shard_range, shard_seeds

    for total, world in [(8, 2), (10, 3), (1 << 23, 8), (5, 8)]:
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    s = np.concatenate([shard_seeds(100, 10, r, 3) for r in range(3)])
    np.testing.assert_array_equal(s, np.arange(100, 110, dtype=np.uint64))  # seeds independent of the sharding


class _FakeEnv:
    def __init__(self, ret):
        self._r = ret

    def episode_returns(self):
        return self._r, None


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ns_gym_amd.distributed import all_gather_returns, shard_range

    res = {}
    for total in (12, 11):   # even and ragged shards
        lo, hi = shard_range(total, rank, world)
        local = torch.arange(lo, hi, dtype=torch.float32) * 0.5
        res[total] = all_gather_returns(_FakeEnv(local)).tolist()
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_returns_gloo_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for total in (12, 11):
        want = [i * 0.5 for i in range(total)]
        assert res[0][total] == want and res[1][total] == want
