"""N>1 path on CPU: shard arithmetic and the obs all-gather over gloo,
world_size 2 (the GPU pools themselves need a GPU; the collective layer and the
id/seed partition do not)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from envpool_amd.sharding import all_gather_rows, shard_range


def test_shard_range_partitions():
    for total in (1, 7, 8, 65536, 262144, 1000003):
        for world in (1, 2, 3, 8):
            if total < world:
                continue
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0
            for (o0, c0), (o1, _) in zip(spans, spans[1:]):
                assert o0 + c0 == o1
            assert spans[-1][0] + spans[-1][1] == total
            counts = [c for _, c in spans]
            assert max(counts) - min(counts) <= 1
    assert shard_range(262144, 3, 8) == (3 * 32768, 32768)
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    off, cnt = shard_range(total, rank, world)
    # fake "obs" of this shard: row i holds its global env id
    ids = torch.arange(off, off + cnt, dtype=torch.float64)
    local = torch.stack([ids, ids * 2, ids + 0.5], dim=1)
    full = all_gather_rows(local, total)
    ok = bool(torch.equal(full[:, 0], torch.arange(total, dtype=torch.float64)))
    ok &= full.shape == (total, 3)
    # weak-scaling bookkeeping of bench.py: max over ranks of the elapsed time
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok &= float(t) == float(world)
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [64, 65])
def test_all_gather_rows_gloo_world2(total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, True), (1, True)]


def _env_worker(rank, world, port, total, steps, q):
    """Each rank steps ITS shard of a `total`-env CartPole pool (the CPU oracle stands in for the device pool: the
    partition rule -- env i of the job is seeded seed + i whatever rank owns it, `info:env_id` global -- is the
    same `env_id_offset` arithmetic), all-gathers the obs / reward / env_id rows every step and compares with the
    single pool of `total` envs stepped with the same actions."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.orc import Oracle

    off, cnt = shard_range(total, rank, world)
    shard = Oracle("CartPole", cnt, seed=7 + off, max_episode_steps=11)  # env j of the shard: seed 7 + off + j
    whole = Oracle("CartPole", total, seed=7, max_episode_steps=11)
    a, w = shard.reset(), whole.reset()
    rng = np.random.default_rng(3)
    ok = True
    for t in range(steps):
        obs = all_gather_rows(torch.from_numpy(a["obs"]), total)
        rew = all_gather_rows(torch.from_numpy(a["reward"]), total)
        ids = all_gather_rows(torch.from_numpy(a["info:env_id"] + off), total)  # the pool adds env_id_offset
        ok &= bool(np.array_equal(obs.numpy(), w["obs"])) and bool(np.array_equal(rew.numpy(), w["reward"]))
        ok &= bool(np.array_equal(ids.numpy().ravel(), np.arange(total)))
        act = rng.integers(0, 2, total).astype(np.int32)  # the same stream on every rank
        a, w = shard.step(act[off:off + cnt]), whole.step(act)
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [32, 33])
def test_sharded_rollout_equals_the_single_pool(total):
    """world_size 2, even and uneven shards, 40 steps with truncations + auto-resets inside: the union of the
    shards IS the single-pool rollout, row for row."""
    from oracle import orc
    if not orc.have_port():
        pytest.skip("oracle/_build/liboracle.so not built")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_env_worker, args=(r, world, port, total, 40, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, True), (1, True)]
