"""Multi-GPU: one process per GPU, envs sharded by contiguous index.

Env instances are independent (own state, θ, t, streams), so there is NO per-step exchange;
seeds are `base + global_index`, which makes results independent of the sharding.  The one
collective of the path is the all-gather of per-env episode returns at rollout end
(f32[N_local] per rank -> f32[world * N_local] on every rank) over RCCL / xGMI.
"""
from __future__ import annotations

import torch


def shard_range(total_envs: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous partition; the first `total % world` ranks take one extra env."""
    base, extra = divmod(int(total_envs), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_seeds(base_seed: int, total_envs: int, rank: int, world: int):
    import numpy as np

    lo, hi = shard_range(total_envs, rank, world)
    return np.arange(lo, hi, dtype=np.uint64) + np.uint64(base_seed)


def all_gather_returns(env, group=None, sizes=None) -> torch.Tensor:
    """All-gather of the last finished episode return of every env (one collective, 4 B/env).  `sizes`: the per-rank
    shard sizes when the caller knows them (e.g. `shard_range` for every rank) - otherwise they are exchanged first."""
    import torch.distributed as dist

    local = env.episode_returns()[0]
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    if sizes is None:
        n_local = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
        all_n = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(all_n, n_local, group=group)
        sizes = [int(x.item()) for x in all_n]
    assert len(sizes) == world and sizes[dist.get_rank(group)] == local.numel()
    if len(set(sizes)) == 1:
        out = torch.empty(world * sizes[0], dtype=local.dtype, device=local.device)
        if dist.get_backend(group) == "gloo":   # gloo has no all_gather_into_tensor for device tensors
            parts = list(out.view(world, sizes[0]).unbind(0))
            dist.all_gather(parts, local.contiguous(), group=group)
        else:
            dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    # ragged shards (total_envs % world != 0): pad to the largest shard, gather, drop the padding
    m = max(sizes)
    padded = torch.zeros(m, dtype=local.dtype, device=local.device)
    padded[: local.numel()] = local
    out = torch.empty(world * m, dtype=local.dtype, device=local.device)
    if dist.get_backend(group) == "gloo":
        dist.all_gather(list(out.view(world, m).unbind(0)), padded, group=group)
    else:
        dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * m: r * m + sizes[r]] for r in range(world)])
