"""Parity of the zero-copy device path (epa_send_device / epa_recv_device, the
path bench.py measures) with the host path (epa_send / epa_recv), whose outputs
are the ones compared with the oracle everywhere else.

The reference's analogue is the XLA custom call (envpool/core/xla.h:116-213):
same Send/Recv, buffers handed over on the accelerator's stream.  Its test
(envpool/atari/atari_envpool_test.py `test_xla`) checks that the XLA results
equal the numpy-API results; this file does the same for the device path,
including partial env_id batches, the documented buffer lifetime and the
producer/consumer stream ordering (epa_send_device's `wait_event`,
epa_wait_stream, epa_consumer_wait).
"""
import numpy as np
import pytest

from envpool_amd.core.device_pool import DevicePool

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

# family, params, max_episode_steps, action sampler
FAMILIES = {
    "HalfCheetah": (dict(), 1000, lambda rng, n: rng.uniform(-1, 1, (n, 6))),
    "Ant": (dict(), 1000, lambda rng, n: rng.uniform(-1, 1, (n, 8))),
    "CartPole": (dict(), 30, lambda rng, n: rng.integers(0, 2, n).astype(np.int32)),
    "FrozenLake": (dict(size=4), 20, lambda rng, n: rng.integers(0, 4, n).astype(np.int32)),
    "Pendulum": (dict(version=1), 25, lambda rng, n: rng.uniform(-2, 2, (n, 1)).astype(np.float32)),
}


def _pools(family, n, seed=7):
    params, max_steps, sampler = FAMILIES[family]
    a = DevicePool(family, n, seed=seed, max_episode_steps=max_steps, params=params)
    b = DevicePool(family, n, seed=seed, max_episode_steps=max_steps, params=params)
    return a, b, sampler


def _host_step(pool, ids, act):
    if act is None:
        pool.reset(ids)
    else:
        pool.send(ids, act)
    return pool.recv_dict()


def _dev_step(pool, ids_t, act_t):
    from envpool_amd.torch_interop import recv_device_tensors, send_device_tensors

    send_device_tensors(pool, act_t, ids_t)
    out = recv_device_tensors(pool)
    return {k: v.cpu().numpy() for k, v in out.items()}


def _assert_same(a, b, ctx):
    assert list(a) == list(b), ctx
    for k in a:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        assert x.dtype == y.dtype and x.shape == y.shape, (ctx, k, x.dtype, y.dtype, x.shape, y.shape)
        assert np.array_equal(x, y, equal_nan=True), (ctx, k, np.abs(x.astype(np.float64) - y).max())


@pytest.mark.parametrize("family", list(FAMILIES))
def test_device_path_bit_identical_to_host_path(family):
    """Same seed, same actions: send/recv and send_device/recv_device give the same bits
    for every state key, through auto-resets (short episodes) and a forced reset."""
    n = 192
    host, dev, sampler = _pools(family, n)
    rng = np.random.default_rng(0)
    ids = np.arange(n, dtype=np.int32)
    device = torch.device("cuda", 0)
    _assert_same(_host_step(host, ids, None), _dev_step(dev, None, None), (family, "reset"))
    for t in range(45):
        act = sampler(rng, n)
        act_t = torch.as_tensor(act, device=device)
        _assert_same(_host_step(host, ids, act), _dev_step(dev, None, act_t), (family, t))
    # forced reset of a subset through the device path (d_action == NULL)
    sub = np.array([5, 3, 100, 64, 191], dtype=np.int32)
    _assert_same(_host_step(host, sub, None),
                 _dev_step(dev, torch.as_tensor(sub, device=device), None), (family, "partial reset"))


@pytest.mark.parametrize("family", ["HalfCheetah", "CartPole", "FrozenLake"])
def test_device_path_partial_env_id_batches(family):
    """A permuted partial d_env_id batch: rows come back in send order and only the listed
    envs advance (sync-mode ordering, state_buffer.h:94-97)."""
    n = 160
    host, dev, sampler = _pools(family, n)
    rng = np.random.default_rng(1)
    device = torch.device("cuda", 0)
    ids = np.arange(n, dtype=np.int32)
    _assert_same(_host_step(host, ids, None), _dev_step(dev, None, None), (family, "reset"))
    for t in range(25):
        k = int(rng.integers(1, n))
        sub = rng.permutation(n)[:k].astype(np.int32)
        act = sampler(rng, k)
        got = _dev_step(dev, torch.as_tensor(sub, device=device), torch.as_tensor(act, device=device))
        want = _host_step(host, sub, act)
        _assert_same(want, got, (family, t, k))
        assert np.array_equal(got["info:env_id"].ravel(), sub)
    # the persistent state of both pools agrees for ALL envs afterwards
    assert np.array_equal(host.get_state(), dev.get_state())


def test_recv_device_buffers_valid_until_second_next_recv():
    """include/envpool_amd.h: pointers stay valid until the SECOND next epa_recv_device
    (batches are double buffered) -- so the batch of step t may still be read after the
    recv of step t+1, and every batch is a distinct buffer from its predecessor."""
    from envpool_amd.torch_interop import recv_device_tensors, send_device_tensors

    n = 4096
    pool = DevicePool("HalfCheetah", n, seed=1, max_episode_steps=1000)
    device = torch.device("cuda", 0)
    gen = torch.Generator(device=device)
    gen.manual_seed(3)
    send_device_tensors(pool, None)
    prev = recv_device_tensors(pool)
    prev_copy = {k: v.clone() for k, v in prev.items()}
    for t in range(6):
        act = torch.rand((n, 6), generator=gen, device=device, dtype=torch.float64) * 2 - 1
        send_device_tensors(pool, act)
        cur = recv_device_tensors(pool)
        torch.cuda.synchronize()
        for k in prev:  # the previous batch is untouched by this step's kernel
            assert torch.equal(prev[k], prev_copy[k]), (t, k)
            assert prev[k].data_ptr() != cur[k].data_ptr()
        prev, prev_copy = cur, {k: v.clone() for k, v in cur.items()}


@pytest.mark.parametrize("how", ["wait_event", "wait_stream"])
def test_send_device_is_ordered_behind_the_action_producer(how):
    """The action batch is written by a slow producer on ANOTHER stream; the step kernel
    must wait for it (epa_send_device's wait_event / epa_wait_stream) without any host
    synchronisation.  Without the ordering the kernel would step with the stale zeros."""
    n = 8192
    device = torch.device("cuda", 0)
    ref = DevicePool("HalfCheetah", n, seed=11, max_episode_steps=1000)
    pool = DevicePool("HalfCheetah", n, seed=11, max_episode_steps=1000)
    ids = np.arange(n, dtype=np.int32)
    ref.reset(ids)
    ref.recv()
    pool.send_device(None)
    pool.recv_device()
    rng = np.random.default_rng(5)
    side = torch.cuda.Stream(device=device)
    buf = torch.zeros((n, 6), device=device, dtype=torch.float64)
    for t in range(4):
        act = rng.uniform(-1, 1, (n, 6))
        src = torch.as_tensor(act, device=device)
        torch.cuda.synchronize()
        buf.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            torch.cuda._sleep(20_000_000)  # ~10 ms of spinning before the actions appear
            buf.copy_(src)
            ev = torch.cuda.Event()
            ev.record(side)
        if how == "wait_event":
            pool.send_device(buf.data_ptr(), wait_event=ev.cuda_event)
        else:
            pool.wait_stream(side.cuda_stream)
            pool.send_device(buf.data_ptr())
        ptrs, k = pool.recv_device()
        # consumer on torch's current stream, ordered behind the step kernel
        pool.consumer_wait(torch.cuda.current_stream(device).cuda_stream)
        from envpool_amd.torch_interop import _DevArray

        key = [kk for kk, _, _ in pool.state_keys].index("obs")
        obs = torch.as_tensor(_DevArray(ptrs[key], (k, 17), np.float64), device=device).clone()
        ref.send(ids, act)
        want = ref.recv_dict()["obs"]
        assert np.array_equal(obs.cpu().numpy(), want), t


def test_empty_send_keeps_the_pool_usable():
    """a k == 0 send/reset enqueues nothing and must not desync recv."""
    n = 64
    # recv_timeout_ms = 0: "nothing pending" is an error instead of a wait (tests/test_gpu_blocking_recv.py)
    pool = DevicePool("CartPole", n, seed=0, max_episode_steps=100, params={"recv_timeout_ms": 0})
    ids = np.arange(n, dtype=np.int32)
    pool.reset(ids)
    pool.recv()
    pool.send(np.zeros((0,), np.int32), np.zeros((0,), np.int32))
    pool.reset(np.zeros((0,), np.int32))
    pool.send(ids, np.zeros(n, np.int32))
    out = pool.recv_dict()
    assert out["obs"].shape == (n, 4)
    with pytest.raises(RuntimeError):
        pool.recv()


def test_step_device_is_send_device_plus_recv_device():
    """epa_step_device (the device path's sync step(): envpool/python/envpool.py:345-349 send, then recv) hands out
    the same batches as the two calls it replaces, bit for bit, and its cached pointer lists follow the blocks."""
    from envpool_amd.torch_interop import _DevArray

    n = 4096
    a = DevicePool("CartPole", n, seed=3, max_episode_steps=50)
    b = DevicePool("CartPole", n, seed=3, max_episode_steps=50)
    acts = [torch.randint(0, 2, (n,), device="cuda", dtype=torch.int32) for _ in range(6)]
    torch.cuda.synchronize()
    a.send_device(None)
    a.recv_device()
    b.step_device(None)

    def tensors(pool, ptrs, k):
        pool.synchronize()
        return {name: torch.as_tensor(_DevArray(p, (k, *shape), dtype), device="cuda").cpu().numpy()
                for (name, dtype, shape), p in zip(pool.state_keys, ptrs)}

    for t in range(60):
        a.send_device(acts[t % 6].data_ptr())
        pa, ka = a.recv_device()
        pb, kb = b.step_device(acts[t % 6].data_ptr())
        assert ka == kb == n
        _assert_same(tensors(a, pa, ka), tensors(b, pb, kb), f"step {t}")
