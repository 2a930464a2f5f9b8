"""recv() BLOCKS like the reference's (AsyncEnvPool::Recv -> StateBufferQueue::Wait sits on a semaphore,
envpool/core/async_envpool.h:169-181, state_buffer_queue.h:148-163, with the GIL released by the binding,
py_envpool.h:255-262): a consumer thread may call recv() BEFORE the producer thread's send() / reset().

Covered: the ctypes path (DevicePool), the Python adaptors (envpool_amd.make), the reference's own pybind11 shim
over the C ABI (integration/refbind), the in-process sharded pool and the Atari pool (host emulator workers);
sync and async mode; and the `recv_timeout_ms` extension key (0 = raise at once for single-threaded callers).
The C++ twin (two std::threads through DeviceEnvPool<Spec>) is integration/refbind/refbind_cc_test.cc.
"""
import os
import queue
import sys
import threading
import time

import numpy as np
import pytest

import envpool_amd as envpool
from envpool_amd.core.device_pool import DevicePool

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


class _Consumer(threading.Thread):
    """Calls `recv` `count` times; keeps every batch; hands each batch to `on_batch`."""

    def __init__(self, recv, count, on_batch=None):
        super().__init__(daemon=True)
        self.recv, self.count, self.on_batch = recv, count, on_batch
        self.batches, self.error = [], None
        self.entered = threading.Event()

    def run(self):
        try:
            for _ in range(self.count):
                self.entered.set()
                b = self.recv()
                self.batches.append(b)
                if self.on_batch:
                    self.on_batch(b)
        except BaseException as e:  # noqa: BLE001 (reported by the test thread)
            self.error = e

    def finish(self, timeout=120):
        self.join(timeout)
        assert not self.is_alive(), "consumer still blocked in recv()"
        if self.error is not None:
            raise self.error


def _blocked(consumer, seconds=0.3):
    """The consumer has entered recv() and has not returned from it."""
    assert consumer.entered.wait(10)
    time.sleep(seconds)
    return consumer.is_alive() and not consumer.batches


@pytest.mark.parametrize("task,adim", [("CartPole", 0), ("HalfCheetah", 6)])
def test_sync_consumer_thread_enters_recv_before_the_producers_send(task, adim):
    """DevicePool (ctypes over the C ABI), sync mode: the consumer's first recv() precedes reset(); then T steps
    with the two threads free-running.  Every batch equals the single-threaded rollout bit for bit, in send order."""
    n, steps = 512, 30
    rng = np.random.default_rng(0)
    acts = [rng.integers(0, 2, n).astype(np.int32) if adim == 0 else rng.uniform(-1, 1, (n, adim))
            for _ in range(steps)]
    ids = np.arange(n, dtype=np.int32)
    ref = DevicePool(task, n, seed=3, max_episode_steps=200)
    ref.reset(ids)
    want = [ref.recv()]
    for a in acts:
        ref.send(ids, a)
        want.append(ref.recv())

    pool = DevicePool(task, n, seed=3, max_episode_steps=200)
    c = _Consumer(pool.recv, steps + 1)
    c.start()
    assert _blocked(c), "recv() returned (or raised) with nothing sent"
    pool.reset(ids)
    for a in acts:
        pool.send(ids, a)  # no waiting for the consumer: batches queue up in send order
    c.finish()
    assert len(c.batches) == steps + 1
    for got, exp in zip(c.batches, want):
        for g, e in zip(got, exp):
            assert np.array_equal(g, e)


def test_async_actor_loop_with_consumer_first():
    """Async mode (batch_size < num_envs): consumer thread in recv() before async_reset(); it hands the env ids of
    each batch to the producer, which sends their next actions (the reference README's actor loop, two threads)."""
    n, b, rounds = 256, 64, 25
    pool = DevicePool("Pendulum", n, batch_size=b, seed=1, max_episode_steps=200, params={"version": 1})
    handed: "queue.Queue[np.ndarray]" = queue.Queue()
    keys = [k[0] for k in pool.state_keys]
    i_id, i_el = keys.index("info:env_id"), keys.index("elapsed_step")
    total = rounds * (n // b) + n // b
    c = _Consumer(pool.recv, total, on_batch=lambda batch: handed.put(batch[i_id].copy()))
    c.start()
    assert _blocked(c)
    pool.reset(np.arange(n, dtype=np.int32))
    for _ in range(rounds * (n // b)):
        ids = handed.get(timeout=60)
        pool.send(ids, np.zeros((len(ids), 1), dtype=np.float32))
    c.finish()
    rows = np.zeros(n, dtype=int)
    for batch in c.batches:
        assert batch[i_id].shape == (b,)
        for e, el in zip(batch[i_id], batch[i_el]):
            assert el == rows[e]  # an env's rows arrive in its own step order (no episode ends in 26 steps)
            rows[e] += 1
    assert (rows == rounds + 1).all()


def test_python_api_recv_blocks_until_async_reset_and_send():
    env = envpool.make_gym("CartPole-v1", num_envs=32, batch_size=8, seed=0)
    c = _Consumer(env.recv, 1)
    c.start()
    assert _blocked(c)
    env.async_reset()
    c.finish()
    obs, rew, term, trunc, info = c.batches[0]
    assert obs.shape == (8, 4) and (info["elapsed_step"] == 0).all()
    # sync pool through the same API: step() from one thread while another one already waits in recv()
    env = envpool.make_gym("CartPole-v1", num_envs=16, seed=0)
    env.reset()
    c = _Consumer(env.recv, 1)
    c.start()
    assert _blocked(c)
    env.send(np.zeros(16, dtype=np.int32))
    c.finish()
    assert (c.batches[0][4]["elapsed_step"] == 1).all()


def test_recv_timeout_key():
    """recv_timeout_ms = 0: RuntimeError at once (single-threaded callers that would otherwise hang);
    > 0: RuntimeError after that long; rows that ARE pending are returned whatever the key says."""
    ids = np.arange(8, dtype=np.int32)
    pool = DevicePool("CartPole", 8, seed=0, params={"recv_timeout_ms": 0})
    with pytest.raises(RuntimeError, match="nothing pending"):
        pool.recv()
    with pytest.raises(RuntimeError, match="nothing pending"):
        pool.recv_device()
    pool.reset(ids)
    assert pool.recv()[0].shape == (8,)
    pool = DevicePool("CartPole", 8, batch_size=4, seed=0, params={"recv_timeout_ms": 200})
    pool.reset(ids[:2])  # fewer rows than a batch holds
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match="only 2 rows pending"):
        pool.recv()
    assert 0.15 < time.perf_counter() - t0 < 5.0
    pool.reset(ids[2:4])
    assert pool.recv()[0].shape == (4,)
    env = envpool.make_gym("CartPole-v1", num_envs=4, seed=0, recv_timeout_ms=0)
    with pytest.raises(RuntimeError, match="nothing pending"):
        env.recv()


def test_producer_is_not_held_up_by_a_waiting_consumer():
    """While the consumer sits in recv() on an EMPTY queue the producer's send / reset / get_state go through
    (mu_ is released while waiting), and a second consumer is serialised behind the first."""
    n = 64
    ids = np.arange(n, dtype=np.int32)
    pool = DevicePool("CartPole", n, seed=2)
    c1, c2 = _Consumer(pool.recv, 1), _Consumer(pool.recv, 1)
    c1.start()
    assert _blocked(c1, 0.2)
    c2.start()
    assert _blocked(c2, 0.2)
    assert pool.get_state().shape == (n, pool.state_dim())  # takes the pool's lock: must not deadlock
    pool.reset(ids)
    pool.send(ids, np.zeros(n, dtype=np.int32))
    c1.finish()
    c2.finish()
    el = sorted(int(c.batches[0][2][0]) for c in (c1, c2))
    assert el == [0, 1]


def test_refbind_pybind_recv_blocks_with_the_gil_released():
    """The reference's own PyEnvPool::PyRecv (py_envpool.h:255-262 releases the GIL) over DeviceEnvPool<Spec>:
    consumer thread first, sync and async."""
    from test_refbind import _refbind

    rb = _refbind()
    for batch in (0, 8):
        spec_cls, pool_cls = rb._CartPoleEnvSpec, rb._CartPoleEnvPool
        conf = dict(zip(spec_cls._config_keys, spec_cls._default_config_values))
        conf.update(num_envs=32, batch_size=batch, seed=4, max_episode_steps=500)
        pool = pool_cls(spec_cls(tuple(conf[k] for k in spec_cls._config_keys)))
        b = batch or 32
        n_batches = 32 // b
        c = _Consumer(pool._recv, n_batches + 1)
        c.start()
        assert _blocked(c)
        pool._reset(np.arange(32, dtype=np.int32))
        while len(c.batches) < n_batches:  # the reset rows
            time.sleep(0.01)
            assert c.error is None
        first = c.batches[0]
        i_id = list(pool_cls._state_keys).index("info:env_id")
        ids = np.asarray(first[i_id], dtype=np.int32)
        pool._send([ids, ids, np.zeros(len(ids), dtype=np.int32)])
        c.finish()
        i_el = list(pool_cls._state_keys).index("elapsed_step")
        assert np.asarray(c.batches[-1][i_el]).tolist() == [1] * b
        assert np.asarray(c.batches[-1][i_id]).tolist() == ids.tolist()


def test_sharded_pool_recv_blocks():
    """device=[0, 0]: the in-process sharded pool splits a send over its shards; recv waits for the split."""
    env = envpool.make_gym("CartPole-v1", num_envs=64, seed=0, device=[0, 0])
    c = _Consumer(env.recv, 2)
    c.start()
    assert _blocked(c)
    env.async_reset()
    env.send(np.zeros(64, dtype=np.int32))
    c.finish()
    assert (c.batches[0][4]["elapsed_step"] == 0).all() and (c.batches[1][4]["elapsed_step"] == 1).all()
    assert c.batches[1][4]["env_id"].tolist() == list(range(64))


@pytest.mark.parametrize("batch_size", [0, 2])
def test_atari_pool_recv_blocks(batch_size):
    """Host-side emulator workers behind the same C ABI: Recv waits for the producer too."""
    from test_gpu_atari_env import make_pool

    pool = make_pool("default", batch_size=batch_size)
    n = pool.num_envs
    b = batch_size or n
    c = _Consumer(pool.recv, n // b)
    c.start()
    assert _blocked(c)
    pool.reset(np.arange(n, dtype=np.int32))
    c.finish()
    got = np.concatenate([batch[0] for batch in c.batches])
    assert sorted(got.tolist()) == list(range(n))
    pool.close()
