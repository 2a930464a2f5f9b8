"""GPU tests of the public API (envpool_amd.make ... step/reset/send/recv),
written like the reference's own tests:
  envpool/classic_control/classic_control_test.py:34-57 (determinism, bounds)
  envpool/toy_text/toy_text_test.py (spaces), envpool/atari/api_test.py:120-330
  docs/content/python_interface.rst:298-327 (auto-reset table)
"""
import numpy as np
import pytest

import envpool_amd as envpool

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("task_id", ["CartPole-v1", "Pendulum-v1", "MountainCar-v0",
                                     "MountainCarContinuous-v0", "Acrobot-v1"])
def test_classic_determinism_and_bounds(task_id):
    """Same seed => identical rollout; different seed => different; obs in space."""
    num_envs = 4
    env0 = envpool.make_gym(task_id, num_envs=num_envs, seed=0)
    env1 = envpool.make_gym(task_id, num_envs=num_envs, seed=0)
    env2 = envpool.make_gym(task_id, num_envs=num_envs, seed=1)
    act_space = env0.action_space
    obs_space = env0.observation_space
    rng = np.random.default_rng(0)
    env0.reset(), env1.reset(), env2.reset()
    eps = np.finfo(np.float32).eps
    differ = False
    for _ in range(2000):
        if hasattr(act_space, "n"):
            action = rng.integers(0, act_space.n, num_envs)
        else:
            action = rng.uniform(act_space.low, act_space.high, (num_envs, *act_space.shape))
        obs0 = env0.step(action)[0]
        obs1 = env1.step(action)[0]
        obs2 = env2.step(action)[0]
        np.testing.assert_array_equal(obs0, obs1)
        differ |= not np.allclose(obs0, obs2)
        assert np.all(obs_space.low - eps <= obs0) and np.all(obs0 <= obs_space.high + eps)
    assert differ


def test_autoreset_table_gym():
    """A step on a finished env resets it and discards the action."""
    env = envpool.make_gym("CartPole-v0", num_envs=2, seed=3, max_episode_steps=5)
    obs, info = env.reset()
    assert info["elapsed_step"].tolist() == [0, 0]
    elapsed = []
    for t in range(14):
        obs, rew, term, trunc, info = env.step(np.array([0, 1]))
        elapsed.append(info["elapsed_step"].tolist())
        done = term | trunc
        if info["elapsed_step"][0] == 0:
            assert rew[0] == 0.0 and not done[0]  # reset row: reward 0
    flat = [e[1] for e in elapsed]
    assert 0 in flat[1:]  # an auto-reset happened
    i = flat.index(0, 1)
    assert flat[i - 1] >= 1 and flat[i + 1] == 1


def test_dm_timestep_semantics():
    env = envpool.make_dm("FrozenLake-v1", num_envs=8, seed=1)
    ts = env.reset()
    assert ts.first().all() and (ts.discount == 1).all() and (ts.reward == 0).all()
    saw_last = False
    for _ in range(300):
        prev = ts
        ts = env.step(np.random.default_rng(0).integers(0, 4, 8).astype(np.int32))
        # after LAST comes FIRST (auto-reset) with discount 1
        assert (ts.step_type[prev.step_type == 2] == 0).all()
        assert (ts.discount[ts.step_type == 2] == 0).all()
        saw_last |= bool((ts.step_type == 2).any())
    assert saw_last
    assert ts.observation.obs.dtype == np.int32


def test_send_recv_async_api():
    env = envpool.make_gym("Pendulum-v1", num_envs=16, batch_size=4, seed=0)
    assert env.is_async
    env.async_reset()
    seen = set()
    for _ in range(20):
        obs, rew, term, trunc, info = env.recv()
        assert obs.shape == (4, 3)
        ids = info["env_id"]
        seen |= set(ids.tolist())
        env.send(np.zeros((4, 1), dtype=np.float32), ids)
    assert seen == set(range(16))


@pytest.mark.parametrize("task,adim,precision", [("HalfCheetah-v4", 6, 64), ("Hopper-v4", 3, 64), ("Ant-v4", 8, 64),
                                                 ("Ant-v4", 8, 32), ("Humanoid-v5", 17, None),
                                                 ("HumanoidStandup-v4", 17, None)])
def test_mujoco_run_to_run_determinism(task, adim, precision):
    """Same seed, same actions => bit-identical rollouts (the reference's
    mujoco_gym_deterministic_test.py:70-123), for every kernel variant."""
    n = 256
    kw = {} if precision is None else {"precision": precision}  # Humanoid: fp64 only
    e0 = envpool.make_gym(task, num_envs=n, seed=5, **kw)
    e1 = envpool.make_gym(task, num_envs=n, seed=5, **kw)
    np.testing.assert_array_equal(e0.reset()[0], e1.reset()[0])
    rng = np.random.default_rng(1)
    for _ in range(40):
        a = rng.uniform(-1, 1, (n, adim))
        r0, r1 = e0.step(a), e1.step(a)
        np.testing.assert_array_equal(r0[0], r1[0])
        np.testing.assert_array_equal(r0[1], r1[1])


def test_planar_fp32_mode_is_gone():
    """precision=32 on the planar families is refused loudly (removed in round 4: outside 1e-5 and
    slower than the fp64 lane-group kernel); Ant keeps its fp32 mode."""
    for task in ("HalfCheetah-v4", "Walker2d-v4", "Hopper-v4"):
        with pytest.raises(ValueError, match="precision must be 64"):
            envpool.make_gym(task, num_envs=4, precision=32)
    from envpool_amd.core.device_pool import DevicePool
    with pytest.raises(Exception, match="precision"):
        DevicePool("HalfCheetah", 4, seed=0, max_episode_steps=10, params={"precision": 0})
    envpool.make_gym("Ant-v4", num_envs=4, precision=32).reset()


def test_halfcheetah_api_shapes_and_determinism():
    n = 256
    env0 = envpool.make("HalfCheetah-v4", "gymnasium", num_envs=n, seed=7)
    env1 = envpool.make("HalfCheetah-v4", "gymnasium", num_envs=n, seed=7)
    o0, i0 = env0.reset()
    o1, _ = env1.reset()
    assert o0.shape == (n, 17) and o0.dtype == np.float64
    np.testing.assert_array_equal(o0, o1)
    rng = np.random.default_rng(0)
    for _ in range(50):
        a = rng.uniform(-1, 1, (n, 6))
        r0, r1 = env0.step(a), env1.step(a)
        np.testing.assert_array_equal(r0[0], r1[0])  # same binary => bit identical
        np.testing.assert_array_equal(r0[1], r1[1])
    obs, rew, term, trunc, info = r0
    assert rew.dtype == np.float32 and rew.shape == (n,)
    assert set(info) >= {"reward_run", "reward_ctrl", "x_position", "x_velocity",
                         "env_id", "elapsed_step"}
    assert info["x_velocity"].dtype == np.float64
    np.testing.assert_allclose(rew, (info["reward_run"] + info["reward_ctrl"]).astype(np.float32),
                               rtol=1e-6)
    with pytest.raises(RuntimeError):
        env0.step(np.zeros((n, 5)))


def test_env_seed_list_matches_offset_seeds():
    a = envpool.make_gym("CartPole-v1", num_envs=4, seed=[10, 11, 12, 13])
    b = envpool.make_gym("CartPole-v1", num_envs=4, seed=10)
    np.testing.assert_array_equal(a.reset()[0], b.reset()[0])


@pytest.mark.parametrize("task,obs_dim,adim", [("HalfCheetah-v4", 17, 6), ("Ant-v4", 27, 8),
                                               ("Humanoid-v4", 376, 17),
                                               ("HumanoidStandup-v5", 348, 17)])
def test_mujoco_frame_stack(task, obs_dim, adim):
    """frame_stack semantics of envpool/mujoco/frame_stack.h:109-135, checked the
    way the reference does (mujoco_gym_envpool_test.cc:58-112): reset replicates
    the frame, a step shifts by one; dynamics equal the unstacked pool."""
    n, S = 16, 4
    flat = envpool.make_gym(task, num_envs=n, seed=3, max_episode_steps=12)
    stk = envpool.make_gym(task, num_envs=n, seed=3, max_episode_steps=12, frame_stack=S)
    assert stk.observation_space.shape == (S, obs_dim)
    o1, _ = flat.reset()
    oS, _ = stk.reset()
    assert oS.shape == (n, S, obs_dim)
    hist = np.repeat(o1[:, None, :], S, axis=1)
    np.testing.assert_array_equal(oS, hist)
    rng = np.random.default_rng(0)
    for t in range(30):
        a = rng.uniform(-1, 1, (n, adim))
        o1, r1, te1, tr1, i1 = flat.step(a)
        oS, rS, teS, trS, iS = stk.step(a)
        np.testing.assert_array_equal(r1, rS)
        fresh = i1["elapsed_step"] == 0  # auto-reset rows replicate the new frame
        hist = np.concatenate([hist[:, 1:], o1[:, None, :]], axis=1)
        hist[fresh] = np.repeat(o1[fresh][:, None, :], S, axis=1)
        np.testing.assert_array_equal(oS, hist)


def test_recv_arrays_own_their_memory_zero_copy():
    """py_envpool.h:40-49 / state_buffer_queue.h:149-163: arrays handed out by recv
    are never overwritten by later steps.  Large batches are views of a pinned
    block (epa_recv_block: one D2H, no host memcpy) that is only recycled once
    every view of it is gone."""
    import gc

    from envpool_amd.core.device_pool import DevicePool

    n = 32768
    pool = DevicePool("CartPole", n, seed=3, max_episode_steps=200)
    ids = np.arange(n, dtype=np.int32)
    pool.reset(ids)
    first = pool.recv()
    assert first[-1].base is not None  # a view into the pinned block
    rng = np.random.default_rng(0)
    held, copies = [first], [[a.copy() for a in first]]
    for t in range(8):  # more batches in flight than the free list holds
        pool.send(ids, rng.integers(0, 2, n).astype(np.int32))
        out = pool.recv()
        held.append(out)
        copies.append([a.copy() for a in out])
    for out, cp in zip(held, copies):
        for a, c in zip(out, cp):
            np.testing.assert_array_equal(a, c)
    # distinct blocks while all are alive
    addrs = {o[0].__array_interface__["data"][0] for o in held}
    assert len(addrs) == len(held)
    del held, out, first
    gc.collect()
    assert sum(len(v) for v in pool._blocks._free.values()) <= pool._blocks._MAX_FREE
    # recycled blocks are reused: steady state allocates nothing new
    pool.send(ids, rng.integers(0, 2, n).astype(np.int32))
    a = pool.recv()
    assert a[0].__array_interface__["data"][0] in addrs


@pytest.mark.parametrize("task,adim", [("HalfCheetah", 6), ("Walker2d", 6), ("Ant", 8),
                                       ("Humanoid", 17), ("HumanoidStandup", 17)])
def test_results_independent_of_batch_composition(task, adim):
    """An env's trajectory must not depend on which other envs share its batch / wave
    (the reference's envs are independent objects): the sync pool (all envs per step)
    and an async pool (batch_size = n / 4, several batches in flight, rows arriving in
    shuffled groups) produce bit-identical observations for every (env, step)."""
    from envpool_amd.core.device_pool import DevicePool

    n, T = 256, 12
    acts = np.random.default_rng(5).uniform(-1, 1, size=(T + 8, n, adim))  # [step, env]
    ids = np.arange(n, dtype=np.int32)
    sync = DevicePool(task, n, seed=11, max_episode_steps=1000)
    sync.reset(ids)
    first = sync.recv_dict()
    ref, ref_el = [first["obs"].copy()], [first["elapsed_step"].ravel().copy()]
    for t in range(T):
        sync.send(ids, acts[t])
        out = sync.recv_dict()
        ref.append(out["obs"].copy())
        ref_el.append(out["elapsed_step"].ravel().copy())  # episodes may end and auto-reset
    ref, ref_el = np.stack(ref), np.stack(ref_el)  # [T + 1, n, ...]

    apool = DevicePool(task, n, batch_size=n // 4, seed=11, max_episode_steps=1000)
    perm = np.random.default_rng(0).permutation(n).astype(np.int32)
    for c in range(4):  # resets submitted in shuffled groups: 4 batches in flight
        apool.reset(perm[c * (n // 4):(c + 1) * (n // 4)])
    count = np.zeros(n, dtype=np.int64)  # rows received so far per env = index of its next row
    checked = 0
    while count.min() <= T:
        out = apool.recv_dict()
        eid = out["info:env_id"].ravel()
        t = count[eid]
        ok = t <= T
        if task.startswith("Humanoid"):
            # the PGS formulation (register-resident A + R vs streamed rows) is chosen per WAVE
            # from its row count, so an env's result may differ in the last bits with the
            # company it keeps (observed 1e-13); everything else about it is identical
            np.testing.assert_allclose(out["obs"][ok], ref[t[ok], eid[ok]], rtol=1e-9, atol=1e-10)
        else:
            np.testing.assert_array_equal(out["obs"][ok], ref[t[ok], eid[ok]])
        np.testing.assert_array_equal(out["elapsed_step"].ravel()[ok], ref_el[t[ok], eid[ok]])
        checked += int(ok.sum())
        count[eid] += 1
        apool.send(eid, acts[np.minimum(t, T + 7), eid])  # every row goes straight back in
    assert checked == n * (T + 1)


@pytest.mark.parametrize("task,adim,horizon", [("HalfCheetah-v4", 6, 1000), ("Humanoid-v4", 17, 300)])
def test_mujoco_full_episode_determinism_and_truncation(task, adim, horizon):
    """The reference's determinism test runs whole episodes
    (mujoco_gym_deterministic_test.py:70-123): same seed => identical rollout up to and
    across the time limit; at elapsed_step == max_episode_steps every surviving env reports
    truncated (not terminated) and the next step returns its reset row."""
    n = 128
    e0 = envpool.make_gym(task, num_envs=n, seed=13, max_episode_steps=horizon)
    e1 = envpool.make_gym(task, num_envs=n, seed=13, max_episode_steps=horizon)
    e2 = envpool.make_gym(task, num_envs=n, seed=14, max_episode_steps=horizon)
    o0, o1, o2 = e0.reset()[0], e1.reset()[0], e2.reset()[0]
    np.testing.assert_array_equal(o0, o1)
    assert not np.array_equal(o0, o2)  # different seed => different rollout
    rng = np.random.default_rng(2)
    lo, hi = e0.action_space.low[0], e0.action_space.high[0]
    for t in range(1, horizon + 2):
        a = rng.uniform(lo, hi, (n, adim))
        r0, r1 = e0.step(a), e1.step(a)
        for x, y in zip(r0[:4], r1[:4]):
            np.testing.assert_array_equal(x, y)
        obs, rew, term, trunc, info = r0
        assert np.isfinite(obs).all() and np.isfinite(rew).all()
        el = info["elapsed_step"]
        assert (trunc == ((el == horizon) & ~term)).all()
        assert not (term & trunc).any()
        if task.startswith("HalfCheetah"):  # never terminates early: one clock for all envs
            assert (el == (t if t <= horizon else 0)).all()
            assert trunc.all() == (t == horizon)


# ---- async mode: independent batches on several compute streams ---------------------------------
def _async_rollout(task, adim, n, b, streams, recvs, device_path, break_rule=False):
    """recv -> send loop of an async pool; returns {env: [(elapsed_step, reward, obs bytes), ...]}.
    The action of (env, its k-th step) is a fixed function, so streams / scheduling cannot matter."""
    from envpool_amd.core.device_pool import DevicePool

    pool = DevicePool(task, n, batch_size=b, seed=3, max_episode_steps=9, params={"compute_streams": streams})
    ids = np.arange(n, dtype=np.int32)
    table = np.random.default_rng(7).uniform(-1, 1, size=(64, n, adim))
    count = np.zeros(n, dtype=np.int64)
    seq = {e: [] for e in range(n)}
    if device_path:
        import torch

        from envpool_amd.torch_interop import recv_device_tensors, send_device_tensors
        dev = torch.device("cuda", 0)
        ttable = torch.as_tensor(table, device=dev)
        tids = torch.arange(n, device=dev, dtype=torch.int32)
        for j in range(n // b):  # recv_device hands out whole launches: n / b reset launches of b rows
            send_device_tensors(pool, None, tids[j * b:(j + 1) * b])
    else:
        pool.reset(ids)
    for r in range(recvs):
        if device_path:
            out = recv_device_tensors(pool)
            eids_t = out["info:env_id"].reshape(-1)  # aliases the handed-out batch: the send continues ITS stream
            if device_path == "clone":
                eids_t = eids_t.clone()  # ids from elsewhere: ordered behind the handed-out batches by events
            host = {k: v.cpu().numpy() for k, v in out.items()}
        else:
            host = pool.recv_dict()
        eids = host["info:env_id"].ravel().astype(np.int64)
        assert len(eids) == b
        for row, e in enumerate(eids):
            seq[int(e)].append((int(host["elapsed_step"][row]), float(host["reward"][row]),
                                host["obs"][row].tobytes()))
        act = table[count[eids] % 64, eids]
        count[eids] += 1
        if device_path:
            act_t = torch.as_tensor(np.ascontiguousarray(act), device=dev)
            send_device_tensors(pool, act_t, eids_t)
        else:
            pool.send(eids.astype(np.int32), act)
            if break_rule and r % 5 == 2:
                # against the rule: the same envs again before they were received (their rows come
                # back twice); must behave as on one stream, where launches simply queue up
                act2 = table[count[eids] % 64, eids]
                count[eids] += 1
                pool.send(eids.astype(np.int32), act2)
    pool.synchronize()
    return seq


@pytest.mark.parametrize("task,adim", [("HalfCheetah", 6), ("Ant", 8)])
@pytest.mark.parametrize("device_path", [False, "lent", "clone"])
def test_async_batches_on_several_streams_match_one_stream(task, adim, device_path):
    """batch_size < num_envs: successive batches run on different compute streams (the reference's workers
    step all queued slices in parallel, async_envpool.h:116-132).  Every env's own sequence of outputs is
    bit-identical to the single-stream run, through auto-resets, on the host and the device path."""
    if device_path:
        pytest.importorskip("torch")
    n, b, recvs = 4096, 512, 120
    one = _async_rollout(task, adim, n, b, 1, recvs, device_path)
    many = _async_rollout(task, adim, n, b, 4, recvs, device_path)
    steps = 0
    for e in range(n):
        assert one[e] == many[e], e
        steps += len(one[e])
    assert steps == recvs * b and max(len(v) for v in one.values()) >= 10


@pytest.mark.parametrize("task", ["Humanoid", "HumanoidStandup"])
def test_async_humanoid_batches_on_several_streams_match_one_stream(task):
    """Round 4: the quad Humanoid kernels keep their per-launch scratch (workspace, cost-sort permutation) per
    compute stream, so an async pool runs their batches concurrently too.  Same batches => same waves => every
    env's sequence bit-identical to the single-stream run (host path and device path)."""
    n, b, recvs = 1024, 256, 48
    for device_path in (False, "lent"):
        one = _async_rollout(task, 17, n, b, 1, recvs, device_path)
        many = _async_rollout(task, 17, n, b, 4, recvs, device_path)
        for e in range(n):
            assert one[e] == many[e], (device_path, e)


def test_async_streams_tolerate_a_resend_before_recv():
    """An env sent again before it was received (the reference would race) is ordered behind its own
    previous step: same per-env sequences as the single-stream engine."""
    n, b, recvs = 2048, 256, 40
    one = _async_rollout("HalfCheetah", 6, n, b, 1, recvs, False, break_rule=True)
    many = _async_rollout("HalfCheetah", 6, n, b, 4, recvs, False, break_rule=True)
    for e in range(n):
        assert one[e] == many[e], e


@pytest.mark.parametrize("streams", [1, 4])
def test_tiny_batches_read_in_place_see_every_rewrite_of_the_staging_slot(streams):
    """Engine key small_zero_copy: the step kernel of a tiny host-path batch reads ids and action rows straight out of
    the pinned staging slot (a ring of three).  Eight sends back to back -- no host synchronisation in between, every
    slot rewritten and re-read several times -- must use the actions of THEIR send: the rollout equals the one of a
    pool that uploads by DMA, bit for bit."""
    from envpool_amd.core.device_pool import DevicePool

    n, b = 512, 64
    pools = [DevicePool("Pendulum", n, batch_size=b, seed=11, max_episode_steps=50,
                        params={"small_zero_copy": z, "compute_streams": streams}) for z in (0, 1)]
    rng = np.random.default_rng(5)
    ids = np.arange(n, dtype=np.int32)
    for p in pools:
        p.reset(ids)
    got = [[p.recv_dict() for _ in range(n // b)] for p in pools]
    for rnd in range(40):
        acts = rng.uniform(-2, 2, size=(n // b, b, 1)).astype(np.float32)
        for p, g in zip(pools, got):
            for j in range(n // b):
                p.send(g[j]["info:env_id"], acts[j])
        got = [[p.recv_dict() for _ in range(n // b)] for p in pools]
        for a, z in zip(got[0], got[1]):
            for k in a:
                assert a[k].tobytes() == z[k].tobytes(), (k, rnd)
