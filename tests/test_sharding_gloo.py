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
