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


def _lsr(z: torch.Tensor, s: int) -> torch.Tensor:
    """Logical right shift of int64 bit patterns (torch's >> is arithmetic)."""
    return (z >> s) & ((1 << (64 - s)) - 1)


def _s64(v: int) -> int:
    """A 64-bit constant as the signed value with the same bit pattern (torch has no uint64 arithmetic)."""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >> 63 else v


def global_actions(k: int, lo: int, hi: int, n_actions: int = 2, device=None, salt: int = 123) -> torch.Tensor:
    """Synthetic random discrete actions of the envs with GLOBAL indices [lo, hi) at step-pool slot `k`: a counter-based
    draw (SplitMix64 of (salt, k, global index)), so what env i is fed does not depend on how the job is sharded - the same
    job run on 1, 2 or 8 ranks steps every env through the same trajectory (tests/test_gpu_sharding_invariance.py)."""
    i = torch.arange(int(lo), int(hi), dtype=torch.int64, device=device)
    z = i + _s64(0x9E3779B97F4A7C15 * (1 + int(k)) + (int(salt) << 40))
    z = (z ^ _lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _s64(0x94D049BB133111EB)
    z = z ^ _lsr(z, 31)
    return (_lsr(z, 33) % int(n_actions)).to(torch.int32)


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


def all_gather_returns_direct(env, group=None, sizes=None) -> torch.Tensor:
    """The same exchange with an explicit DIRECT schedule: every rank sends its shard to each of the other ranks and receives
    theirs, all 2 x (world - 1) transfers posted as one batch of point-to-point operations (RCCL groups them into one launch).
    On MI355X the 8 GPUs of a node are a full xGMI mesh - 7 links per GPU, each to one peer - so this moves every shard over its
    own link once (4 MiB per link at 2^20 envs per GPU), where a ring all-gather forwards every shard through 7 hops of single
    links.  Whether it beats RCCL's own choice is a measurement (bench.py times both)."""
    import torch.distributed as dist

    local = env.episode_returns()[0].contiguous()
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if sizes is None:
        n_local = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
        all_n = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(all_n, n_local, group=group)
        sizes = [int(x.item()) for x in all_n]
    assert len(sizes) == world and sizes[rank] == local.numel()
    offs = [0]
    for s in sizes:
        offs.append(offs[-1] + s)
    out = torch.empty(offs[-1], dtype=local.dtype, device=local.device)
    out[offs[rank]:offs[rank + 1]] = local
    host = dist.get_backend(group) == "gloo"     # the CPU rehearsal backend moves host memory
    src = local.cpu() if host else local
    parts = {r: (torch.empty(sizes[r], dtype=local.dtype) if host else out[offs[r]:offs[r + 1]]) for r in range(world) if r != rank}
    ops = []
    for d in range(1, world):                     # peer order staggered by rank: at any moment every link carries one transfer
        to, frm = (rank + d) % world, (rank - d) % world
        ops.append(dist.P2POp(dist.isend, src, to, group=group))
        ops.append(dist.P2POp(dist.irecv, parts[frm], frm, group=group))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    if host:
        for r, p in parts.items():
            out[offs[r]:offs[r + 1]] = p.to(out.device)
    return out


def verify_gather(gathered: torch.Tensor, local: torch.Tensor, lo: int, group=None) -> bool:
    """What an all-gather of returns must satisfy, checked on every rank: this rank's slice of the gathered tensor IS its local
    tensor, and the sum over ranks of each rank's local checksum equals the checksum of the gathered tensor (bit patterns summed as
    integers: exact, order-independent).  Returns the AND over ranks."""
    import torch.distributed as dist

    ok = bool(torch.equal(gathered[lo:lo + local.numel()], local))
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        bits = lambda t: t.contiguous().view(torch.int32).to(torch.int64).sum()   # noqa: E731
        v = torch.stack([bits(local), torch.tensor(0 if ok else 1, dtype=torch.int64, device=local.device)])
        if dist.get_backend(group) == "gloo":
            v = v.cpu()
        dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
        ok = int(v[1]) == 0 and int(v[0]) == int(bits(gathered))
    return ok
