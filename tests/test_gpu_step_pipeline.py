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
