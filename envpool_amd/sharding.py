"""Multi-GPU sharding of one logical pool.

Envs are independent, so a pool of `num_envs_total` envs shards embarrassingly:
rank g of G (one process per GPU, `torch.distributed` over RCCL) owns the
contiguous id range [offset, offset + count).  Seeds and `info:env_id` stay
global (`epa_config.env_id_offset`), so the union of all shards reproduces the
single-pool rollout exactly.  No collective is on the data path; a learner that
wants every rank to hold the whole observation batch all-gathers the obs
section (`all_gather_rows`), a direct one-hop exchange over xGMI.
"""

from __future__ import annotations

from typing import Any


def shard_range(num_envs_total: int, rank: int, world_size: int) -> tuple[int, int]:
    """(offset, count) of rank's shard: contiguous, sizes differ by at most 1."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(num_envs_total, world_size)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def make_shard(family: str, num_envs_total: int, rank: int, world_size: int,
               device: int | None = None, **kwargs: Any) -> Any:
    """DevicePool holding this rank's shard of a `num_envs_total`-env pool."""
    from envpool_amd.core.device_pool import DevicePool

    offset, count = shard_range(num_envs_total, rank, world_size)
    env_seed = kwargs.pop("env_seed", None)
    if env_seed is not None:
        env_seed = list(env_seed)[offset:offset + count]
    return DevicePool(family, count, device=rank if device is None else device,
                      env_id_offset=offset, env_seed=env_seed, **kwargs)


def all_gather_rows(local: Any, num_envs_total: int, group: Any = None) -> Any:
    """All-gather a per-shard `[count, ...]` tensor into `[num_envs_total, ...]`
    ordered by env id.  Works with the RCCL ("nccl") and gloo backends; shards
    may differ in size by one row."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = [shard_range(num_envs_total, r, world)[1] for r in range(world)]
    assert local.shape[0] == counts[rank], (local.shape, counts[rank])
    if len(set(counts)) == 1:
        out = torch.empty((num_envs_total, *local.shape[1:]), dtype=local.dtype,
                          device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    # uneven shards: pad to the largest shard, gather, drop the padding rows
    maxc = max(counts)
    padded = torch.zeros((maxc, *local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: counts[rank]] = local
    out = torch.empty((world * maxc, *local.shape[1:]), dtype=local.dtype,
                      device=local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    out = out.view(world, maxc, *local.shape[1:])
    return torch.cat([out[r, : counts[r]] for r in range(world)], dim=0)
