"""The pipelined sync step of the host path (Pool::SendPipelined, engine key "step_pipeline"): a whole-pool step of a
sync pool runs as two launches over the two halves of the rows, the first half's download under the second launch.
Everything recv returns must be bit-identical to the single-launch step, rows in send order
(envpool/python/envpool.py:345-349 send-then-recv; state_buffer.h:94-97 row order)."""
import numpy as np
import pytest

from envpool_amd.core.device_pool import DevicePool

pytestmark = pytest.mark.gpu


def _actions(rng, task, n):
    if task == "CartPole":
        return rng.integers(0, 2, n).astype(np.int32)
    adim = {"HalfCheetah": 6, "Walker2d": 6, "Ant": 8}[task]
    return rng.uniform(-1, 1, (n, adim))


@pytest.mark.parametrize("task,n,steps", [("HalfCheetah", 65536, 6), ("HalfCheetah", 40000, 4), ("Walker2d", 36864, 4),
                                          ("Ant", 32768, 3), ("CartPole", 131072, 30)])
def test_pipelined_step_is_bit_identical_to_the_single_launch_step(task, n, steps):
    ids = np.arange(n, dtype=np.int32)
    one = DevicePool(task, n, seed=11, max_episode_steps=20, params={"step_pipeline": 0})
    two = DevicePool(task, n, seed=11, max_episode_steps=20, params={"step_pipeline": 32768})
    one.reset(ids), two.reset(ids)
    a, b = one.recv(), two.recv()
    rng = np.random.default_rng(3)
    keys = [k for k, _, _ in one.state_keys]
    for t in range(steps):
        for name, x, y in zip(keys, a, b):
            assert x.shape == y.shape and np.array_equal(x, y), (task, t, name)
        assert np.array_equal(b[0], ids)  # info:env_id: rows in send order
        act = _actions(rng, task, n)
        one.send(ids, act), two.send(ids, act)
        a, b = one.recv(), two.recv()
    for name, x, y in zip(keys, a, b):
        assert np.array_equal(x, y), (task, "last", name)


def test_pipelined_steps_interleave_with_partial_sends_and_resets():
    """A partial send, a reset of some envs and the device path between pipelined steps: same results as without."""
    n = 65536
    ids = np.arange(n, dtype=np.int32)
    pools = [DevicePool("HalfCheetah", n, seed=5, max_episode_steps=1000, params={"step_pipeline": sp}) for sp in (0, 32768)]
    rng = np.random.default_rng(0)
    acts = [rng.uniform(-1, 1, (n, 6)) for _ in range(4)]
    some = rng.permutation(n)[:5000].astype(np.int32)
    outs = []
    for p in pools:
        seq = []
        p.reset(ids)
        seq.append(p.recv())
        p.send(ids, acts[0])
        seq.append(p.recv())
        p.send(some, acts[1][some])           # partial send: one launch
        seq.append(p.recv())
        p.reset(some[:100])
        seq.append(p.recv())
        p.send(ids, acts[2]), p.send(ids, acts[3])  # two steps queued before a recv
        seq.append(p.recv()), seq.append(p.recv())
        outs.append(seq)
    for x, y in zip(*outs):
        for u, v in zip(x, y):
            assert np.array_equal(u, v)


# ---- the DIRECT step (Pool::SendInto / epa_send_into): results written by the step kernel into the block named at send time
@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("task,n,adim", [("Hopper", 20000, 3), ("Ant", 9000, 8), ("HalfCheetah", 20000, 6),
                                         ("HalfCheetah", 65536, 6), ("Pusher", 8192, 7), ("Humanoid", 2048, 17),
                                         ("CartPole", 300000, 0)])
def test_direct_step_is_bit_identical_and_rows_keep_their_memory(task, n, adim, mode):
    """`DevicePool.send` of a whole sync pool names a pinned block; with "direct_out" the step kernel writes the rows
    straight into it.  Same bytes as the download path over steps, auto-resets, a partial send, a forced reset and two
    steps queued before a recv; and the arrays of an earlier step are never overwritten by a later one
    (py_envpool.h:40-49: every batch owns its memory)."""
    ids = np.arange(n, dtype=np.int32)
    # (mode 2: the action rows are read in place out of the pinned staging slot too; it goes in front of the
    # two-launch pipeline, mode 1 behind it)
    pools = [DevicePool(task, n, seed=13, max_episode_steps=6, params={"direct_out": d}) for d in (0, mode)]
    rng = np.random.default_rng(8)

    def act(k):
        return rng.integers(0, 2, k).astype(np.int32) if adim == 0 else rng.uniform(-1, 1, (k, adim))

    acts = [act(n) for _ in range(9)]
    some = rng.permutation(n)[: n // 7].astype(np.int32)
    seqs = []
    for p in pools:
        seq = []
        p.reset(ids)
        seq.append(p.recv())
        for t in range(4):
            p.send(ids, acts[t])
            seq.append(p.recv())
        p.send(some, acts[4][some])  # partial: an ordinary send
        seq.append(p.recv())
        p.reset(some[:50])
        seq.append(p.recv())
        p.send(ids, acts[5]), p.send(ids, acts[6])  # two blocks posted before a recv
        seq.append(p.recv()), seq.append(p.recv())
        p.send(ids, acts[7])
        seq.append(p.recv())
        seqs.append(seq)
    kept = [[x.copy() for x in batch] for batch in seqs[1]]
    for t, (a, b) in enumerate(zip(*seqs)):
        for (name, _, _), x, y in zip(pools[0].state_keys, a, b):
            assert x.shape == y.shape and x.tobytes() == y.tobytes(), (task, t, name)
    pools[1].send(ids, acts[8])  # one more step: nothing handed out before may change
    pools[1].recv()
    for batch, copy in zip(seqs[1], kept):
        for x, y in zip(batch, copy):
            assert x.tobytes() == y.tobytes()


def test_direct_step_with_the_consumer_already_waiting_in_recv():
    """A consumer thread sits in recv() (with a block of its own) before the producer's send names another block: the
    rows are copied out of the posted block -- same rows, no deadlock."""
    import threading

    n = 20000
    ids = np.arange(n, dtype=np.int32)
    pool = DevicePool("Hopper", n, seed=2, max_episode_steps=1000, params={"direct_out": 2})
    ref = DevicePool("Hopper", n, seed=2, max_episode_steps=1000, params={"direct_out": 0})
    rng = np.random.default_rng(1)
    for p in (pool, ref):
        p.reset(ids)
        p.recv()
    for t in range(3):
        a = rng.uniform(-1, 1, (n, 3))
        got = {}
        th = threading.Thread(target=lambda: got.setdefault("out", pool.recv()))
        th.start()
        import time
        time.sleep(0.05)  # the consumer is inside epa_recv_block by now
        pool.send(ids, a)
        th.join(timeout=30)
        assert not th.is_alive()
        ref.send(ids, a)
        want = ref.recv()
        for x, y in zip(got["out"], want):
            assert x.tobytes() == y.tobytes(), t


@pytest.mark.parametrize("task,adim,streams", [("HalfCheetah", 6, 4), ("Hopper", 3, 1), ("Pusher", 7, 2)])
def test_direct_batches_of_an_async_pool(task, adim, streams):
    """Async mode (batch_size < num_envs): every send of batch_size rows -- the env ids of the last recv, in whatever order
    they came -- names its block; the batch's kernel writes into it on whichever compute stream it runs.  The recv /
    send loop of the reference benchmark (benchmark/test_envpool.py:94-105) gives the same bytes with and without."""
    n, b = 24576, 4096
    pools = [DevicePool(task, n, batch_size=b, seed=21, max_episode_steps=9,
                        params={"direct_out": d, "compute_streams": streams}) for d in (0, 2)]
    rng = np.random.default_rng(4)
    ids = np.arange(n, dtype=np.int32)
    for p in pools:
        p.reset(ids)
    for t in range(40):
        a, z = pools[0].recv_dict(), pools[1].recv_dict()
        for k in a:
            assert a[k].tobytes() == z[k].tobytes(), (k, t)
        act = rng.uniform(-1, 1, (b, adim))
        for p, out in zip(pools, (a, z)):
            p.send(out["info:env_id"], act)
