"""Parity at the headline size (BASELINE.json: num_envs = 65536), where the oracle cannot follow
every env: size-independent properties instead --
  * the last 256 envs of the big pool against the oracle seeded `seed + 65280` (env i of a pool is
    seeded seed + i, envpool/core/env.h:101-110), teacher forced;
  * run-to-run determinism of the whole batch;
  * batch-composition independence: the same 256 envs inside the 65536-env launch and alone in a
    256-env pool (env_id_offset) give the same trajectories;
  * bookkeeping of every row (elapsed_step, done / trunc at max_episode_steps, auto-reset)."""
import numpy as np
import pytest

from envpool_amd.core.device_pool import DevicePool
from oracle.orc import Oracle

pytestmark = pytest.mark.gpu

N = 65536
TAIL = 256


def _reset(pool):
    pool.reset(np.arange(pool.num_envs, dtype=np.int32) + pool.env_id_offset)
    return pool.recv_dict()


def _step(pool, act):
    pool.send(np.arange(pool.num_envs, dtype=np.int32) + pool.env_id_offset, act)
    return pool.recv_dict()


@pytest.mark.parametrize("task,adim,amax,exact", [("HalfCheetah", 6, 1.0, True), ("Humanoid", 17, 0.4, False),
                                                   ("HumanoidStandup", 17, 0.4, False)])
def test_headline_size_properties(task, adim, amax, exact):
    steps, max_steps = 6, 4  # episodes end (truncation) inside the run: auto-reset at full size
    rng = np.random.default_rng(0)
    acts = rng.uniform(-amax, amax, size=(steps, N, adim))
    # the registered -v4 ids run with post_constraint=False (gym/registration.py), which is also
    # the oracle's default
    params = {"post_constraint": 0} if task.startswith("Humanoid") else None
    big = DevicePool(task, N, seed=7, max_episode_steps=max_steps, params=params)
    twin = DevicePool(task, N, seed=7, max_episode_steps=max_steps, params=params)
    # (HalfCheetah: a pool picks its lane layout -- 2 or 4 lanes per env -- from its own size; the
    # small pool is given the big pool's, the two sum the contact rows in different orders)
    small = DevicePool(task, TAIL, seed=7, max_episode_steps=max_steps, env_id_offset=N - TAIL,
                       params={"planar_layout": 2} if task == "HalfCheetah" else params)
    orc = Oracle(task, TAIL, seed=7 + N - TAIL, max_episode_steps=max_steps)
    a, t, s, o = _reset(big), _reset(twin), _reset(small), orc.reset()
    for k in range(steps + 1):
        tail = slice(N - TAIL, N)
        # determinism of the whole batch
        for key in a:
            np.testing.assert_array_equal(a[key], t[key], err_msg=f"{key}@{k}")
        # the same envs alone in a small pool: identical (HalfCheetah) / to rounding (Humanoid, whose
        # PGS formulation is chosen per wave)
        assert np.array_equal(s["info:env_id"].ravel(), np.arange(N - TAIL, N))
        if exact:
            np.testing.assert_array_equal(a["obs"][tail], s["obs"])
        else:
            np.testing.assert_allclose(a["obs"][tail], s["obs"], rtol=1e-9, atol=1e-10)
        # against the oracle
        np.testing.assert_allclose(a["obs"][tail], o["obs"], rtol=1e-7, atol=1e-8, err_msg=f"step {k}")
        for key in ("done", "trunc", "elapsed_step", "step_type"):
            np.testing.assert_array_equal(a[key].ravel()[tail], o[key].ravel(), err_msg=f"{key}@{k}")
        # bookkeeping of every row: step k of an episode of length max_steps, then a reset row
        want = k % (max_steps + 1)
        assert (a["elapsed_step"].ravel() == want).all()
        assert (a["done"].ravel() == (want == max_steps)).all() or task == "Humanoid"  # (Humanoid may fall earlier)
        assert (a["trunc"].ravel() <= a["done"].ravel()).all()
        assert np.isfinite(a["obs"]).all()
        assert np.array_equal(a["info:env_id"].ravel(), np.arange(N))
        if k == steps:
            break
        a, t = _step(big, acts[k]), _step(twin, acts[k])
        s, o = _step(small, acts[k][tail]), orc.step(acts[k][tail])


# ---------------------------------------------------------------------------
# BASELINE.json config 4: Ant-v4, 262144 envs over 8 GPUs = 32768 envs per GPU
# ---------------------------------------------------------------------------
def test_ant_config4_shard_size():
    """The Ant quad kernel at the per-GPU shard size of config 4 (N = 32768): the last 256 envs
    teacher forced against the oracle (obs rtol 1e-9 / atol 1e-10, the bar of
    test_ant_teacher_forced_step), the whole batch run-to-run deterministic, and the same 256
    envs alone in a 256-env pool (env_id_offset) bit-identical to their rows of the big launch.
    max_episode_steps = 5 puts a truncation + auto-reset of every row inside the run."""
    n, steps, max_steps = 32768, 9, 5
    rng = np.random.default_rng(1)
    acts = rng.uniform(-1, 1, size=(steps, n, 8))
    big = DevicePool("Ant", n, seed=7, max_episode_steps=max_steps)
    twin = DevicePool("Ant", n, seed=7, max_episode_steps=max_steps)
    small = DevicePool("Ant", TAIL, seed=7, max_episode_steps=max_steps, env_id_offset=n - TAIL)
    orc = Oracle("Ant", TAIL, seed=7 + n - TAIL, max_episode_steps=max_steps)
    tail = slice(n - TAIL, n)
    tail_ids = np.arange(n - TAIL, n, dtype=np.int32)
    a, t, s, o = _reset(big), _reset(twin), _reset(small), orc.reset()
    worst = 0.0
    for k in range(steps + 1):
        for key in a:
            np.testing.assert_array_equal(a[key], t[key], err_msg=f"{key}@{k}")
        for key in s:
            np.testing.assert_array_equal(a[key][tail], s[key], err_msg=f"small pool {key}@{k}")
        np.testing.assert_allclose(a["obs"][tail], o["obs"], rtol=1e-9, atol=1e-10, err_msg=f"step {k}")
        worst = max(worst, float(np.abs(a["obs"][tail] - o["obs"]).max()))
        for key in ("done", "trunc", "elapsed_step", "step_type"):
            np.testing.assert_array_equal(a[key].ravel()[tail], o[key].ravel(), err_msg=f"{key}@{k}")
        np.testing.assert_allclose(a["reward"].ravel()[tail], o["reward"].ravel(), rtol=1e-6, atol=1e-6)
        assert (a["elapsed_step"].ravel() <= max_steps).all()
        assert (a["trunc"].ravel() <= a["done"].ravel()).all()
        assert np.isfinite(a["obs"]).all()
        assert np.array_equal(a["info:env_id"].ravel(), np.arange(n))
        if k == steps:
            break
        st = orc.get_state()
        big.set_state(st, tail_ids), twin.set_state(st, tail_ids), small.set_state(st)
        a, t = _step(big, acts[k]), _step(twin, acts[k])
        s, o = _step(small, acts[k][tail]), orc.step(acts[k][tail])
    print(f"Ant N={n}: worst teacher-forced |d obs| over the tail = {worst:.3e}")


# ---------------------------------------------------------------------------
# BASELINE.json config 3: HalfCheetah-v4, 8192 envs on one GPU.  Run in fp64 (the reference's
# mjtNum; BASELINE's "fp32" wording predates the removal of the planar fp32 mode).
# ---------------------------------------------------------------------------
def test_halfcheetah_config3_every_env():
    """All 8192 envs of config 3 against the oracle, teacher forced, fp64: obs rtol 1e-9 / atol
    1e-10 (north star: 1e-5 relative), bookkeeping exact."""
    n, steps = 8192, 12
    pool = DevicePool("HalfCheetah", n, seed=13, max_episode_steps=1000, params={"precision": 1})
    orc = Oracle("HalfCheetah", n, seed=13, max_episode_steps=1000)
    a, b = _reset(pool), orc.reset()
    rng = np.random.default_rng(5)
    worst = 0.0
    for t in range(steps):
        np.testing.assert_allclose(a["obs"], b["obs"], rtol=1e-9, atol=1e-10, err_msg=f"step {t}")
        for key in ("done", "trunc", "elapsed_step", "step_type", "info:env_id"):
            np.testing.assert_array_equal(a[key].ravel(), b[key].ravel(), err_msg=f"{key}@{t}")
        worst = max(worst, float(np.abs(a["obs"] - b["obs"]).max()))
        pool.set_state(orc.get_state())
        act = rng.uniform(-1, 1, size=(n, 6))
        a, b = _step(pool, act), orc.step(act)
    print(f"HalfCheetah N={n} fp64: worst teacher-forced |d obs| = {worst:.3e}")


# ---------------------------------------------------------------------------
# BASELINE.json config 2: classic control (and a toy_text id) at 65536 envs on one GPU
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["CartPole-v1", "Pendulum-v1", "Acrobot-v1", "FrozenLake-v1"])
def test_classic_config2_size(name):
    """The last 256 envs of a 65536-env pool against the reference compiled in place
    (oracle/_ref; the restatement it is pinned to where that library did not travel), seeded
    seed + 65280: integer keys exact, float keys within 1e-5 relative over the first 50 free-running
    steps (the horizon of test_classic_golden_rollout; FrozenLake: everything bit-exact, 300 steps);
    the whole batch run-to-run deterministic, env ids and elapsed_step of every row."""
    from hip_util import make_hip_pool
    from oracle.orc import have_ref
    from oracle_cases import CASES, INTEGER_EXACT, sample_actions

    c = CASES[name]
    exact = name in INTEGER_EXACT
    steps = 300 if exact else 50
    big, twin = make_hip_pool(name, N, 3), make_hip_pool(name, N, 3)
    orc = Oracle(c["task"], TAIL, seed=3 + N - TAIL, max_episode_steps=c["max_steps"],
                 extra=c["extra"], kind="reference" if have_ref() else "port")
    tail = slice(N - TAIL, N)
    rng = np.random.default_rng(4)
    a, t, o = _reset(big), _reset(twin), orc.reset()
    n_done = 0
    for k in range(steps + 1):
        for key in a:
            np.testing.assert_array_equal(a[key], t[key], err_msg=f"{key}@{k}")
        for key, want in o.items():
            got = a[key].reshape(N, -1)[tail]
            if want.dtype == np.float32 and not exact:
                np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6, err_msg=f"{name}:{key}@{k}")
            elif key in ("info:env_id", "info:players.env_id"):
                assert np.array_equal(got.ravel(), np.arange(N - TAIL, N))
            else:
                assert np.array_equal(got, want), f"{name}:{key}@{k}"
        assert np.array_equal(a["info:env_id"].ravel(), np.arange(N))
        assert (a["trunc"].ravel() <= a["done"].ravel()).all()
        n_done += int(a["done"].sum())
        if k == steps:
            break
        act = sample_actions(c, rng, N)
        a, t, o = _step(big, act), _step(twin, act), orc.step(act[tail])
    if name in ("CartPole-v1", "FrozenLake-v1"):  # the others rarely end an episode within the horizon
        assert n_done > 0  # auto-resets happened at full size
    print(f"{name} N={N} vs oracle kind={orc.kind}: {steps} steps, {n_done} episode ends")


# ---------------------------------------------------------------------------
# Headline-size parity on MID-EPISODE states (feet down, joint limits active): the oracle follows a
# random rollout of the tail envs for 400 steps; at steps 200, 240, ... 400 its state is forced
# into the tail rows of the full-size launch and one env-step of both is compared at the
# teacher-forced bar (1e-9).  The reset-adjacent test above never sees these states.
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("task,n,adim,params", [
    ("HalfCheetah", 65536, 6, {"precision": 1}),
    ("Walker2d", 65536, 6, {"precision": 1}),
    ("Ant", 32768, 8, None),
])
def test_full_size_mid_episode_states(task, n, adim, params):
    rng = np.random.default_rng(5)
    big = DevicePool(task, n, seed=3, max_episode_steps=1000, params=params)
    orc = Oracle(task, TAIL, seed=3 + n - TAIL, max_episode_steps=1000)
    tail = slice(n - TAIL, n)
    tail_ids = np.arange(n - TAIL, n, dtype=np.int32)
    _reset(big), orc.reset()
    for _ in range(3):  # the other rows leave their reset states too
        _step(big, rng.uniform(-1, 1, size=(n, adim)))
    worst, checked, contacts = 0.0, 0, 0
    for t in range(1, 401):
        act_tail = rng.uniform(-1, 1, size=(TAIL, adim))
        if t >= 200 and t % 40 == 0:
            st = orc.get_state()
            big.set_state(st, tail_ids)
            act = rng.uniform(-1, 1, size=(n, adim))
            act[tail] = act_tail
            a, o = _step(big, act), orc.step(act_tail)
            # rows whose env was done are RESET rows: fresh mt19937 draws, and the two generators are not
            # in step here (set_state carries the physics state, not the generator) -- every other row
            # is an env-step from the forced state
            live = o["elapsed_step"].ravel() > 0
            np.testing.assert_allclose(a["obs"][tail][live], o["obs"][live], rtol=1e-9, atol=1e-10,
                                       err_msg=f"{task} t={t}")
            for key in ("done", "trunc", "elapsed_step", "step_type"):
                np.testing.assert_array_equal(a[key].ravel()[tail], o[key].ravel(), err_msg=f"{key}@{t}")
            np.testing.assert_allclose(a["reward"].ravel()[tail], o["reward"].ravel(), rtol=1e-6, atol=1e-6)
            worst = max(worst, float(np.abs(a["obs"][tail][live] - o["obs"][live]).max()))
            checked += 1
            # the states are mid-episode ones: most envs are past step 20 of their episode
            contacts += int((o["elapsed_step"].ravel() > 20).sum())
            assert np.isfinite(a["obs"]).all()
        else:
            orc.step(act_tail)
    assert checked == 6 and contacts > 0
    print(f"{task} N={n}: worst teacher-forced |d obs| on mid-episode states = {worst:.3e} "
          f"({contacts} env-steps past step 20 of their episode)")
