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


@pytest.mark.parametrize("task,adim,amax,exact", [("HalfCheetah", 6, 1.0, True), ("Humanoid", 17, 0.4, False)])
def test_headline_size_properties(task, adim, amax, exact):
    steps, max_steps = 6, 4  # episodes end (truncation) inside the run: auto-reset at full size
    rng = np.random.default_rng(0)
    acts = rng.uniform(-amax, amax, size=(steps, N, adim))
    # the registered -v4 ids run with post_constraint=False (gym/registration.py), which is also
    # the oracle's default
    params = {"post_constraint": 0} if task == "Humanoid" else None
    big = DevicePool(task, N, seed=7, max_episode_steps=max_steps, params=params)
    twin = DevicePool(task, N, seed=7, max_episode_steps=max_steps, params=params)
    small = DevicePool(task, TAIL, seed=7, max_episode_steps=max_steps, env_id_offset=N - TAIL,
                       params=params)
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
        assert (a["done"].ravel() == (want == max_steps)).all() or task == "Humanoid"
        assert (a["trunc"].ravel() <= a["done"].ravel()).all()
        assert np.isfinite(a["obs"]).all()
        assert np.array_equal(a["info:env_id"].ravel(), np.arange(N))
        if k == steps:
            break
        a, t = _step(big, acts[k]), _step(twin, acts[k])
        s, o = _step(small, acts[k][tail]), orc.step(acts[k][tail])
