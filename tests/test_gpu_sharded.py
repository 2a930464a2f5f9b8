"""In-process sharding over several devices (`device=[...]`, SURVEY §8e "host gather")
and the `env_id_offset` shard key, exercised on ONE GPU by listing it twice: the two
shards are real, independent DevicePools with their own streams and host threads, so
everything except the physical second GPU is covered (the reference's analogue of
running several pools side by side is benchmark/numa_test.sh:15-21).
"""
import numpy as np
import pytest

import envpool_amd as envpool

pytestmark = pytest.mark.gpu


def _sample(space, rng, n):
    if hasattr(space, "n"):
        return rng.integers(0, space.n, n).astype(np.int32)
    return rng.uniform(space.low, space.high, (n, *space.shape)).astype(space.dtype)


def _same(a, b, ctx):
    if isinstance(a, dict):
        assert set(a) == set(b), ctx
        for k in a:
            _same(a[k], b[k], (ctx, k))
        return
    a, b = np.asarray(a), np.asarray(b)
    assert a.dtype == b.dtype and a.shape == b.shape, (ctx, a.dtype, b.dtype, a.shape, b.shape)
    assert np.array_equal(a, b, equal_nan=True), ctx


@pytest.mark.parametrize("task_id", ["CartPole-v1", "FrozenLake-v1", "HalfCheetah-v4", "Ant-v4", "Pusher-v4",
                                     "Humanoid-v4"])
def test_two_shards_equal_one_pool(task_id):
    """step()/reset() through device=[0, 0] returns the same rows, in the same order, as a
    single pool with the same seed (env i is seeded seed + i on whichever shard owns it)."""
    n = 128
    one = envpool.make_gym(task_id, num_envs=n, seed=5)
    two = envpool.make_gym(task_id, num_envs=n, seed=5, device=[0, 0])
    rng = np.random.default_rng(0)
    o1, i1 = one.reset()
    o2, i2 = two.reset()
    _same(o1, o2, "reset obs")
    _same(i1, i2, "reset info")
    for t in range(40):
        act = _sample(one.action_space, rng, n)
        r1 = one.step(act)
        r2 = two.step(act)
        for x, y, name in zip(r1, r2, ("obs", "rew", "term", "trunc", "info")):
            _same(x, y, (task_id, t, name))


@pytest.mark.parametrize("task_id", ["CartPole-v1", "HalfCheetah-v4"])
def test_two_shards_partial_and_permuted_ids(task_id):
    """Partial batches: ascending ids (landed directly, no scatter) and a permutation that
    interleaves the shards (scatter fallback) both return rows in SEND order."""
    n = 96
    one = envpool.make_gym(task_id, num_envs=n, seed=9)
    two = envpool.make_gym(task_id, num_envs=n, seed=9, device=[0, 0])
    rng = np.random.default_rng(2)
    one.reset()
    two.reset()
    for t in range(30):
        k = int(rng.integers(1, n + 1))
        ids = rng.permutation(n)[:k].astype(np.int32)
        if t % 2 == 0:
            ids = np.sort(ids)
        act = _sample(one.action_space, rng, k)
        r1 = one.step(act, ids)
        r2 = two.step(act, ids)
        for x, y, name in zip(r1, r2, ("obs", "rew", "term", "trunc", "info")):
            _same(x, y, (task_id, t, name))
        assert np.array_equal(r2[4]["env_id"], ids)
    # partial reset through both
    ids = np.array([70, 3, 48, 47], dtype=np.int32)
    _same(one.reset(ids)[0], two.reset(ids)[0], "partial reset")


def test_recv_arrays_of_sharded_pool_are_never_overwritten():
    """Ownership rule of the reference (py_envpool.h:40-49): arrays handed out by recv are
    not touched by later steps -- also when two shards land in one shared block."""
    n = 32768  # large enough for the pinned-block path
    env = envpool.make_gym("HalfCheetah-v4", num_envs=n, seed=1, device=[0, 0])
    env.reset()
    rng = np.random.default_rng(0)
    obs_a = env.step(rng.uniform(-1, 1, (n, 6)))[0]
    keep = obs_a.copy()
    for _ in range(4):
        env.step(rng.uniform(-1, 1, (n, 6)))
    assert np.array_equal(obs_a, keep)


@pytest.mark.parametrize("task_id", ["CartPole-v1", "HalfCheetah-v4"])
def test_env_id_offset_through_the_host_api(task_id):
    """make(..., env_id_offset=K) is one shard of a bigger pool; reset()/step()/
    async_reset()/reset_mask must address envs by their GLOBAL ids and reproduce rows
    [K, K + n) of the big pool."""
    n, off = 48, 80
    big = envpool.make_gym(task_id, num_envs=off + n, seed=3)
    shard = envpool.make_gym(task_id, num_envs=n, seed=3, env_id_offset=off)
    assert shard.all_env_ids.tolist() == list(range(off, off + n))
    rng = np.random.default_rng(4)
    ob, ib = big.reset()
    os_, is_ = shard.reset()
    _same(ob[off:], os_, "reset obs")
    assert is_["env_id"].tolist() == list(range(off, off + n))
    for t in range(25):
        act = _sample(big.action_space, rng, off + n)
        rb = big.step(act)
        rs = shard.step(act[off:])
        _same(rb[0][off:], rs[0], (t, "obs"))
        _same(rb[1][off:], rs[1], (t, "rew"))
        assert rs[4]["env_id"].tolist() == list(range(off, off + n))
    # explicit global ids and the gymnasium reset_mask
    ids = np.array([off + 5, off + 1], dtype=np.int32)
    act = _sample(big.action_space, rng, 2)
    _same(big.step(act, ids)[0], shard.step(act, ids)[0], "partial step")
    mask = np.zeros(n, dtype=bool)
    mask[[2, 7]] = True
    o, info = shard.reset(options={"reset_mask": mask})
    assert info["env_id"].tolist() == [off + 2, off + 7]
    with pytest.raises(ValueError):
        shard.step(act, np.array([0, 1], dtype=np.int32))  # local ids are out of range
