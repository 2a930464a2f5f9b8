"""GPU parity of the HalfCheetah HIP kernel (through the C ABI) against the
fp64 oracle (oracle/mjcpu — parity of that oracle vs real MuJoCo is UNPINNED,
see oracle/mjcpu/mjcpu.h).

Tolerances (stated): teacher-forced per env-step (5 mj_steps):
  fp64 kernel (the only one since round 4): obs rtol 1e-9 / atol 1e-10
  (the fp32 arithmetic mode of rounds 1-3 -- |d obs| p99 1.6e-4 on a cond~1e4 Newton system, outside
   the 1e-5 bar, and slower than the fp64 lane-group kernel -- was removed)
"""
import os

import numpy as np
import pytest

from envpool_amd.core.device_pool import DevicePool
from oracle.orc import Oracle

pytestmark = pytest.mark.gpu


def make_pair(n, seed, precision, max_steps=1000):
    pool = DevicePool("HalfCheetah", n, seed=seed, max_episode_steps=max_steps,
                      params={"precision": precision})
    orc = Oracle("HalfCheetah", n, seed=seed, max_episode_steps=max_steps)
    return pool, orc


def hip_reset(pool):
    pool.reset(np.arange(pool.num_envs, dtype=np.int32))
    return pool.recv_dict()


def hip_step(pool, act):
    pool.send(np.arange(pool.num_envs, dtype=np.int32), act)
    return pool.recv_dict()


@pytest.mark.parametrize("precision", [1])
def test_reset_matches_oracle(precision):
    pool, orc = make_pair(64, 5, precision)
    a, b = hip_reset(pool), orc.reset()
    # uniform draws are bit exact; normals go through device log/sqrt
    np.testing.assert_array_equal(a["obs"][:, :8], b["obs"][:, :8])
    np.testing.assert_allclose(a["obs"][:, 8:], b["obs"][:, 8:], rtol=1e-14, atol=1e-16)
    for k in ("elapsed_step", "done", "reward", "discount", "step_type", "trunc",
              "info:env_id"):
        np.testing.assert_array_equal(a[k].ravel(), b[k].ravel())


@pytest.mark.parametrize("precision,rtol,atol", [(1, 1e-9, 1e-10)])
def test_teacher_forced_step(precision, rtol, atol):
    n, steps = 256, 120
    pool, orc = make_pair(n, 9, precision)
    hip_reset(pool), orc.reset()
    rng = np.random.default_rng(3)
    worst = 0.0
    for t in range(steps):
        pool.set_state(orc.get_state())
        act = rng.uniform(-1.2, 1.2, size=(n, 6))
        a, b = hip_step(pool, act), orc.step(act)
        np.testing.assert_allclose(a["obs"], b["obs"], rtol=rtol, atol=atol,
                                   err_msg=f"step {t}")
        np.testing.assert_allclose(a["reward"].ravel(), b["reward"].ravel(),
                                   rtol=max(rtol, 1e-6), atol=max(atol * 20, 1e-6))
        for k in ("info:x_position", "info:x_velocity", "info:reward_ctrl"):
            np.testing.assert_allclose(a[k].ravel(), b[k].ravel(), rtol=rtol,
                                       atol=atol * 20)
        worst = max(worst, float(np.abs(a["obs"] - b["obs"]).max()))
    print(f"precision={precision}: worst teacher-forced |d obs| = {worst:.3e}")


@pytest.mark.parametrize("precision", [1])
def test_free_running_horizon(precision):
    n, steps = 128, 300
    pool, orc = make_pair(n, 1, precision)
    a, b = hip_reset(pool), orc.reset()
    rng = np.random.default_rng(7)
    first = None
    for t in range(steps):
        act = rng.uniform(-1, 1, size=(n, 6))
        a, b = hip_step(pool, act), orc.step(act)
        rel = np.abs(a["obs"] - b["obs"]) / (1e-2 + np.abs(b["obs"]))
        if first is None and rel.max() > 1e-5:
            first = t
    print(f"precision={precision}: first step with rel obs error > 1e-5: {first} "
          f"(final max rel {rel.max():.2e})")
    # free-running rollouts are chaotic once contacts switch: the
    # 1e-11 per-step difference of the fp64 kernel grows ~1.4x per step, so only
    # a horizon is asserted; teacher-forced parity is the actual bar.
    if precision == 1:
        assert first is None or first >= 20


def test_episode_bookkeeping_and_autoreset():
    """max_episode_steps=7: done/trunc/step_type/elapsed_step and the
    auto-reset on the following step follow the reference exactly."""
    n = 32
    pool, orc = make_pair(n, 2, 1, max_steps=7)
    a, b = hip_reset(pool), orc.reset()
    rng = np.random.default_rng(0)
    for t in range(30):
        act = rng.uniform(-1, 1, size=(n, 6))
        a, b = hip_step(pool, act), orc.step(act)
        for k in ("elapsed_step", "done", "discount", "step_type", "trunc",
                  "info:env_id"):
            np.testing.assert_array_equal(a[k].ravel(), b[k].ravel(), err_msg=f"{k}@{t}")
    assert b["elapsed_step"].max() <= 7


def test_long_rollout_stays_bounded():
    """The kernel must stay finite and bounded over a long random rollout."""
    n = 1024
    pool = DevicePool("HalfCheetah", n, seed=0, max_episode_steps=1000)
    hip_reset(pool)
    rng = np.random.default_rng(0)
    for t in range(400):
        d = hip_step(pool, rng.uniform(-1, 1, size=(n, 6)))
    assert np.isfinite(d["obs"]).all()
    assert np.abs(d["obs"][:, 8:]).max() < 100


# ---------------------------------------------------------------------------
# Ant-v4 (3-D, free joint, RK4, sphere/capsule contacts)
# ---------------------------------------------------------------------------
def make_ant_pair(n, seed, precision, max_steps=1000):
    pool = DevicePool("Ant", n, seed=seed, max_episode_steps=max_steps,
                      params={"precision": precision})
    orc = Oracle("Ant", n, seed=seed, max_episode_steps=max_steps)
    return pool, orc


def test_ant_reset_matches_oracle():
    pool, orc = make_ant_pair(64, 5, 1)
    a, b = hip_reset(pool), orc.reset()
    np.testing.assert_allclose(a["obs"], b["obs"], rtol=1e-14, atol=1e-16)
    for k in ("elapsed_step", "done", "reward", "step_type", "trunc", "info:env_id"):
        np.testing.assert_array_equal(a[k].ravel(), b[k].ravel())


def test_ant_teacher_forced_fp32_distribution():
    n, steps = 128, 60
    pool, orc = make_ant_pair(n, 4, 0)
    hip_reset(pool), orc.reset()
    rng = np.random.default_rng(3)
    errs = []
    for t in range(steps):
        pool.set_state(orc.get_state())
        act = rng.uniform(-1, 1, size=(n, 8))
        a, b = hip_step(pool, act), orc.step(act)
        live = b["elapsed_step"].ravel() > 0
        errs.append(np.abs(a["obs"] - b["obs"]).max(axis=1)[live])
    errs = np.concatenate(errs)
    med, p99, mx = np.median(errs), np.percentile(errs, 99), errs.max()
    print(f"Ant fp32 teacher-forced |d obs|: median {med:.2e} p99 {p99:.2e} max {mx:.2e}")
    # the max is not asserted: Ant contacts (solimp d0=0.9, margin 0.01) switch on
    # discontinuously, so a sphere whose distance rounds across the margin in fp32
    # gives an O(1) different step for that env (rare: < 1e-3 of env-steps).
    assert med <= 2e-5 and p99 <= 2e-3


@pytest.mark.parametrize("precision,rtol,atol", [(1, 1e-9, 1e-10)])
def test_ant_teacher_forced_step(precision, rtol, atol):
    n, steps = 128, 60
    pool, orc = make_ant_pair(n, 4, precision)
    hip_reset(pool), orc.reset()
    rng = np.random.default_rng(3)
    worst = 0.0
    for t in range(steps):
        pool.set_state(orc.get_state())
        act = rng.uniform(-1, 1, size=(n, 8))
        a, b = hip_step(pool, act), orc.step(act)
        np.testing.assert_allclose(a["obs"], b["obs"], rtol=rtol, atol=atol, err_msg=f"step {t}")
        for k in ("info:x_position", "info:y_position", "info:x_velocity",
                  "info:reward_survive", "info:distance_from_origin"):
            np.testing.assert_allclose(a[k].ravel(), b[k].ravel(), rtol=rtol, atol=atol * 20)
        np.testing.assert_allclose(a["reward"].ravel(), b["reward"].ravel(),
                                   rtol=max(rtol, 1e-6), atol=max(atol * 20, 1e-6))
        for k in ("done", "trunc", "elapsed_step", "step_type"):
            np.testing.assert_array_equal(a[k].ravel(), b[k].ravel(), err_msg=f"{k}@{t}")
        worst = max(worst, float(np.abs(a["obs"] - b["obs"]).max()))
    print(f"Ant precision={precision}: worst teacher-forced |d obs| = {worst:.3e}")


def test_ant_unhealthy_termination_and_autoreset():
    """terminate_when_unhealthy: envs fall over / leave z in [0.2, 1] and reset."""
    n = 64
    pool, orc = make_ant_pair(n, 11, 1, max_steps=40)
    a, b = hip_reset(pool), orc.reset()
    rng = np.random.default_rng(1)
    seen_term = False
    for t in range(80):
        pool.set_state(orc.get_state())
        act = rng.uniform(-1, 1, size=(n, 8))
        a, b = hip_step(pool, act), orc.step(act)
        for k in ("done", "trunc", "elapsed_step", "step_type", "discount"):
            np.testing.assert_array_equal(a[k].ravel(), b[k].ravel(), err_msg=f"{k}@{t}")
        seen_term |= bool((b["done"] & ~b["trunc"]).any())
    assert b["elapsed_step"].max() <= 40


# ---- Walker2d (same planar kernel: mirrored hinges, RK4, healthy termination) ----
def make_walker_pair(n, seed, precision, task="Walker2d", max_steps=1000):
    pool = DevicePool("Walker2d", n, seed=seed, max_episode_steps=max_steps,
                      params={"precision": precision,
                              "xml_v5": 1 if task == "Walker2dV5" else 0,
                              "legacy_healthy_reward": 0 if task == "Walker2dV5" else 1})
    orc = Oracle(task, n, seed=seed, max_episode_steps=max_steps)
    return pool, orc


@pytest.mark.parametrize("precision", [1])
def test_walker_reset_matches_oracle(precision):
    pool, orc = make_walker_pair(64, 5, precision)
    a, b = hip_reset(pool), orc.reset()
    # both qpos and qvel noise are uniform draws (walker2d.h:119-126): bit exact
    np.testing.assert_array_equal(a["obs"], b["obs"])
    assert list(a.keys()) == list(b.keys())
    for k in ("elapsed_step", "done", "reward", "discount", "step_type", "trunc",
              "info:env_id", "info:x_position", "info:x_velocity"):
        np.testing.assert_array_equal(a[k].ravel(), b[k].ravel())


@pytest.mark.parametrize("task", ["Walker2d", "Walker2dV5"])
def test_walker_teacher_forced_step(task):
    """fp64 kernel, 4 RK4 mj_steps (16 forward evaluations) per env-step:
    obs rtol 1e-9 / atol 1e-10; bookkeeping (healthy termination, auto-reset) exact."""
    n, steps = 256, 100
    pool, orc = make_walker_pair(n, 9, 1, task)
    hip_reset(pool), orc.reset()
    rng = np.random.default_rng(3)
    worst, seen_term = 0.0, False
    for t in range(steps):
        pool.set_state(orc.get_state())
        act = rng.uniform(-1.2, 1.2, size=(n, 6))
        a, b = hip_step(pool, act), orc.step(act)
        np.testing.assert_allclose(a["obs"], b["obs"], rtol=1e-9, atol=1e-10,
                                   err_msg=f"step {t}")
        np.testing.assert_allclose(a["reward"].ravel(), b["reward"].ravel(),
                                   rtol=1e-6, atol=1e-6)
        for k in ("info:x_position", "info:x_velocity"):
            np.testing.assert_allclose(a[k].ravel(), b[k].ravel(), rtol=1e-9, atol=2e-9)
        for k in ("done", "trunc", "elapsed_step", "step_type", "discount"):
            np.testing.assert_array_equal(a[k].ravel(), b[k].ravel(), err_msg=f"{k}@{t}")
        seen_term |= bool((b["done"] & ~b["trunc"]).any())
        worst = max(worst, float(np.abs(a["obs"] - b["obs"]).max()))
    print(f"{task} fp64: worst teacher-forced |d obs| = {worst:.3e}")
    assert seen_term  # random actions make the walker fall within ~30 steps


def test_walker_deterministic():
    n = 512
    outs = []
    for _ in range(2):
        pool = DevicePool("Walker2d", n, seed=3, max_episode_steps=1000, params={"precision": 1})
        hip_reset(pool)
        rng = np.random.default_rng(5)
        for t in range(40):
            a = hip_step(pool, rng.uniform(-1, 1, size=(n, 6)))
        outs.append(a["obs"].copy())
    np.testing.assert_array_equal(outs[0], outs[1])


# ---- InvertedPendulum / InvertedDoublePendulum (mj_pendulum.hip.h: no contacts, RK4) ----
@pytest.mark.parametrize("task,nv,amax,params,extra", [
    ("InvertedPendulum", 2, 3.0, {}, ()),
    ("InvertedDoublePendulum", 3, 1.0, {}, ()),
    # the v5 registrations: reward only while alive, one constraint-force observation
    ("InvertedDoublePendulum", 3, 1.0, {"constraint_obs_dim": 1, "reward_if_not_terminated": 1},
     (5, 0, 0, 0.1, 0, 0, 0, 0, -1, 0, 1, 1)),
])
def test_pendulum_matches_oracle(task, nv, amax, params, extra):
    """Reset (bit-exact uniform draws; device log/sqrt for the normal ones) and
    free-running + limit-pushing teacher-forced steps: obs rtol 1e-9 / atol 1e-10,
    reward and bookkeeping (healthy termination, auto-reset) exact."""
    n = 512
    pool = DevicePool(task, n, seed=4, max_episode_steps=1000, params=params)
    orc = Oracle(task, n, seed=4, max_episode_steps=1000, extra=extra)
    a, b = hip_reset(pool), orc.reset()
    assert list(a.keys()) == list(b.keys())
    np.testing.assert_allclose(a["obs"], b["obs"], rtol=1e-14, atol=1e-16)
    rng = np.random.default_rng(8)
    seen_term, seen_force, worst = False, False, 0.0
    for t in range(120):
        st = orc.get_state()
        if t % 3 == 2:  # push a third of the envs against the joint limits
            sel = rng.random(n) < 0.33
            side = rng.choice([-1.0, 1.0], n)
            st[sel, 0] = (side * rng.uniform(0.97, 1.005, n))[sel]
            st[sel, nv] = (side * rng.uniform(0, 3, n))[sel]
            if nv == 2:
                st[sel, 1] = (side * rng.uniform(1.5, 1.58, n))[sel]
            orc.set_state(st)
        pool.set_state(st)
        act = rng.uniform(-1.2 * amax, 1.2 * amax, size=(n, 1))
        a, b = hip_step(pool, act), orc.step(act)
        np.testing.assert_allclose(a["obs"], b["obs"], rtol=1e-9, atol=1e-10, err_msg=f"step {t}")
        np.testing.assert_allclose(a["reward"].ravel(), b["reward"].ravel(), rtol=1e-6, atol=1e-6)
        for k in ("done", "trunc", "elapsed_step", "step_type", "discount"):
            np.testing.assert_array_equal(a[k].ravel(), b[k].ravel(), err_msg=f"{k}@{t}")
        seen_term |= bool((b["done"] & ~b["trunc"]).any())
        if nv == 3:
            seen_force |= bool((np.abs(b["obs"][:, 8]) > 0).any())
        worst = max(worst, float(np.abs(a["obs"] - b["obs"]).max()))
    print(f"{task}: worst teacher-forced |d obs| = {worst:.3e}")
    assert seen_term and (nv == 2 or seen_force)


# ---- Ant-v3 / Ant-v5: contact-force observations (cfrc_ext) ----
@pytest.mark.parametrize("name,params,extra,nobs", [
    # v3: use_contact_force without mj_rnePostConstraint => cfrc_ext stays zero
    ("Ant-v3", {"use_contact_force": 1},
     (5, 0.5, 1.0, 0.1, 0, 0, 0, 0, -1, 0, 0, 3, 1, 0, 0, -1), 111),
    # v5: cfrc_ext of the last forward evaluation, world body excluded
    ("Ant-v5", {"use_contact_force": 1, "post_constraint": 1,
                "exclude_worldbody_contact_forces": 1, "legacy_healthy_reward": 0},
     (5, 0.5, 1.0, 0.1, 0, 0, 0, 0, -1, 0, 0, 3, 1, 1, 1, 0), 105),
])
def test_ant_contact_force_observation(name, params, extra, nobs):
    n, steps = 128, 60
    pool = DevicePool("Ant", n, seed=6, max_episode_steps=1000, params={"precision": 1, **params})
    orc = Oracle("Ant", n, seed=6, max_episode_steps=1000, extra=extra)
    a, b = hip_reset(pool), orc.reset()
    assert a["obs"].shape == (n, nobs) == b["obs"].shape
    np.testing.assert_array_equal(a["obs"][:, 27:], 0.0)  # mj_resetData zeros cfrc_ext
    rng = np.random.default_rng(2)
    nz, cost = 0, 0.0
    for t in range(steps):
        pool.set_state(orc.get_state())
        act = rng.uniform(-1, 1, size=(n, 8))
        a, b = hip_step(pool, act), orc.step(act)
        np.testing.assert_allclose(a["obs"], b["obs"], rtol=1e-8, atol=1e-9, err_msg=f"step {t}")
        np.testing.assert_allclose(a["reward"].ravel(), b["reward"].ravel(), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(a["info:reward_contact"].ravel(),
                                   b["info:reward_contact"].ravel(), rtol=1e-8, atol=1e-12)
        for k in ("done", "trunc", "elapsed_step", "step_type"):
            np.testing.assert_array_equal(a[k].ravel(), b[k].ravel(), err_msg=f"{k}@{t}")
        nz += int((np.abs(b["obs"][:, 27:]) > 0).sum())
        cost = min(cost, float(b["info:reward_contact"].min()))
    if name == "Ant-v3":
        assert nz == 0 and cost == 0.0
    else:
        assert nz > 1000 and cost < 0.0  # real forces were compared


# ---- Reacher (chain kernel without cart; lagged fingertip position; goal rejection sampling) ----
@pytest.mark.parametrize("name,params,extra,nobs", [
    ("Reacher-v4", {}, (), 11),
    ("Reacher-v5", {"reward_after_step": 1, "obs_include_z_distance": 0},
     (2, 1.0, 0, 0, 0, 0, 0, 0, -1, 0, 0, 3, 0, 0, 0, -1, 1, 0), 10),
])
def test_reacher_matches_oracle(name, params, extra, nobs):
    n = 512
    pool = DevicePool("Reacher", n, seed=4, max_episode_steps=50, params=params)
    orc = Oracle("Reacher", n, seed=4, max_episode_steps=50, extra=extra)
    a, b = hip_reset(pool), orc.reset()
    assert list(a.keys()) == list(b.keys()) and a["obs"].shape == (n, nobs)
    # reset: all draws are uniform (incl. the goal's rejection loop) => exact qpos / qvel
    np.testing.assert_allclose(a["obs"], b["obs"], rtol=1e-14, atol=1e-16)
    rng = np.random.default_rng(8)
    worst = 0.0
    # free running across several 50-step episodes (auto-resets redraw goals); the arm
    # has armature 1 >> link inertia, so trajectories do not diverge chaotically
    for t in range(130):
        if t % 10 == 9:  # push joint1 against its +-3 rad limit
            st = orc.get_state()
            sel = rng.random(n) < 0.3
            side = rng.choice([-1.0, 1.0], n)
            st[sel, 1] = (side * rng.uniform(2.9, 3.05, n))[sel]
            st[sel, 5] = (side * rng.uniform(0, 5, n))[sel]
            orc.set_state(st)
            pool.set_state(st)
        act = rng.uniform(-1.2, 1.2, size=(n, 2))
        a, b = hip_step(pool, act), orc.step(act)
        np.testing.assert_allclose(a["obs"], b["obs"], rtol=1e-9, atol=1e-10, err_msg=f"step {t}")
        np.testing.assert_allclose(a["reward"].ravel(), b["reward"].ravel(), rtol=1e-6, atol=1e-6)
        for k in ("info:reward_dist", "info:reward_ctrl"):
            np.testing.assert_allclose(a[k].ravel(), b[k].ravel(), rtol=1e-9, atol=1e-10)
        for k in ("done", "trunc", "elapsed_step", "step_type", "discount"):
            np.testing.assert_array_equal(a[k].ravel(), b[k].ravel(), err_msg=f"{k}@{t}")
        worst = max(worst, float(np.abs(a["obs"] - b["obs"]).max()))
    print(f"{name}: worst free-running |d obs| over 130 steps = {worst:.3e}")


# ---- Pusher (7-dof arm + sliding cylinder; capsule/sphere-cylinder and table contacts) ----
_PUSHER_LO = np.array([-2.2854, -0.5236, -1.5, -2.3213, -1.5, -1.094, -1.5])
_PUSHER_HI = np.array([1.714602, 1.3963, 1.7, 0.0, 1.5, 0.0, 1.5])


@pytest.mark.parametrize("name,task,params,extra", [
    ("Pusher-v4", "Pusher", {}, ()),
    ("Pusher-v5", "PusherV5", {"xml_v5": 1, "reward_after_step": 1, "weighted_reward_info": 1},
     (5, 0.1, 0, 0, 0, 0, 0, 0, -1, 0, 0, 3, 0, 0, 0, -1, 1, 0, 0, 0, 1.0, 0.5, 1)),
])
def test_pusher_matches_oracle(name, task, params, extra):
    n, nq, nv = 512, 11, 11
    pool = DevicePool("Pusher", n, seed=4, max_episode_steps=100, params=params)
    orc = Oracle(task, n, seed=4, max_episode_steps=100, extra=extra)
    a, b = hip_reset(pool), orc.reset()
    assert list(a.keys()) == list(b.keys()) and a["obs"].shape == (n, 23)
    # reset: uniform draws only (cylinder rejection loop, qvel noise) => exact
    np.testing.assert_allclose(a["obs"], b["obs"], rtol=1e-14, atol=1e-16)
    rng = np.random.default_rng(8)
    worst = pushed = 0

    def compare(act, tag):
        nonlocal worst
        a, b = hip_step(pool, act), orc.step(act)
        np.testing.assert_allclose(a["obs"], b["obs"], rtol=1e-9, atol=1e-10, err_msg=tag)
        np.testing.assert_allclose(a["reward"].ravel(), b["reward"].ravel(), rtol=1e-6, atol=1e-6)
        for k in ("info:reward_dist", "info:reward_ctrl", "info:reward_near"):
            np.testing.assert_allclose(a[k].ravel(), b[k].ravel(), rtol=1e-9, atol=1e-10,
                                       err_msg=f"{k} {tag}")
        for k in ("done", "trunc", "elapsed_step", "step_type", "discount"):
            np.testing.assert_array_equal(a[k].ravel(), b[k].ravel(), err_msg=f"{k} {tag}")
        worst = max(worst, float(np.abs(a["obs"] - b["obs"]).max()))
        return b

    for t in range(45):
        # (1) teacher-forced arm pose low over the table (the wrist / forearm capsules reach
        #     the table plane and the cylinder's height), some joints beyond their limits
        st = orc.get_state()
        q = rng.uniform(_PUSHER_LO * 0.6, _PUSHER_HI * 0.6, (n, 7))
        q[:, 1] = rng.uniform(0.25, 0.75, n)
        q[:, 3] = rng.uniform(-0.8, 0.0, n)
        over = rng.random(n) < 0.15
        q[over, 5] = rng.choice([-1.094 - 0.02, 0.02], int(over.sum()))
        st[:, :7] = q
        st[:, nq:nq + 7] = rng.normal(0, 0.5, (n, 7))
        st[:, nq + 7:nq + 9] = 0
        st[:, nq + nv:nq + 2 * nv] = 0
        orc.set_state(st)
        pool.set_state(st)
        b = compare(rng.uniform(-2, 2, (n, 7)), f"pose step {t}")
        # (2) put the cylinder within reach of where the fingertips now are, give it a small
        #     velocity, and step: the arm pushes the cylinder / the cylinder slides on the table
        st = orc.get_state()
        live = b["elapsed_step"].ravel() > 0
        ang, dist = rng.uniform(0, 2 * np.pi, n), rng.uniform(0.0, 0.2, n)
        ox, oy = st[:, -5] + dist * np.cos(ang), st[:, -4] + dist * np.sin(ang)
        st[:, 7], st[:, 8] = oy + 0.05, ox - 0.45    # obj_slidey, obj_slidex (body at 0.45 -0.05)
        st[:, nq + 7:nq + 9] = rng.normal(0, 0.05, (n, 2))
        st[:, -2], st[:, -1] = ox, oy
        orc.set_state(st)
        pool.set_state(st)
        v0 = st[:, nq + 7:nq + 9].copy()
        compare(rng.uniform(-2, 2, (n, 7)), f"push step {t}")
        dv = np.abs(orc.get_state()[:, nq + 7:nq + 9] - v0).max(axis=1)
        pushed += int((dv[live] > 1e-2).sum())
    assert pushed > 200, pushed    # the contact-rich branch really ran
    print(f"{name}: worst teacher-forced |d obs| = {worst:.3e}; steps where the cylinder was "
          f"pushed: {pushed}")


def test_pusher_free_running_episodes():
    """Free-running v4 episodes across auto-resets (100-step truncation); fp64 is the only
    Pusher build."""
    n = 256
    pool = DevicePool("Pusher", n, seed=11, max_episode_steps=100)
    orc = Oracle("Pusher", n, seed=11, max_episode_steps=100)
    hip_reset(pool), orc.reset()
    rng = np.random.default_rng(2)
    for t in range(230):
        act = rng.uniform(-2, 2, (n, 7))
        a, b = hip_step(pool, act), orc.step(act)
        np.testing.assert_allclose(a["obs"], b["obs"], rtol=1e-7, atol=1e-8, err_msg=f"step {t}")
        for k in ("done", "trunc", "elapsed_step"):
            np.testing.assert_array_equal(a[k].ravel(), b[k].ravel(), err_msg=f"{k}@{t}")


# ---- Swimmer (planar floating chain + inertia-box fluid forces) ----
def test_swimmer_matches_oracle():
    n = 512
    pool = DevicePool("Swimmer", n, seed=4, max_episode_steps=1000)
    orc = Oracle("Swimmer", n, seed=4, max_episode_steps=1000)
    a, b = hip_reset(pool), orc.reset()
    assert list(a.keys()) == list(b.keys()) and a["obs"].shape == (n, 8)
    np.testing.assert_array_equal(a["obs"], b["obs"])  # uniform draws only: bit exact
    rng = np.random.default_rng(8)
    worst, hit = 0.0, False
    for t in range(150):
        st = orc.get_state()
        if t % 10 == 9:  # push the two motor hinges against their +-100 deg limits
            sel = rng.random(n) < 0.3
            side = rng.choice([-1.0, 1.0], n)
            st[sel, 3] = (side * rng.uniform(1.70, 1.78, n))[sel]
            st[sel, 8] = (side * rng.uniform(0, 3, n))[sel]
            orc.set_state(st)
        pool.set_state(st)
        act = rng.uniform(-1.2, 1.2, size=(n, 2))
        a, b = hip_step(pool, act), orc.step(act)
        np.testing.assert_allclose(a["obs"], b["obs"], rtol=1e-9, atol=1e-10, err_msg=f"step {t}")
        np.testing.assert_allclose(a["reward"].ravel(), b["reward"].ravel(), rtol=1e-6, atol=1e-6)
        for k in ("info:reward_fwd", "info:reward_ctrl", "info:x_position", "info:y_position",
                  "info:distance_from_origin", "info:x_velocity", "info:y_velocity"):
            np.testing.assert_allclose(a[k].ravel(), b[k].ravel(), rtol=1e-9, atol=2e-9, err_msg=k)
        for k in ("done", "trunc", "elapsed_step", "step_type", "discount"):
            np.testing.assert_array_equal(a[k].ravel(), b[k].ravel(), err_msg=f"{k}@{t}")
        hit |= bool((np.abs(b["obs"][:, 1:3]) > 1.74).any())
        worst = max(worst, float(np.abs(a["obs"] - b["obs"]).max()))
    print(f"Swimmer: worst teacher-forced |d obs| = {worst:.3e}")
    assert hit  # the hinge limit rows were exercised


# ---- Hopper (planar kernel, ghost second leg, margin 0.001, body-body capsule contacts) ----
def make_hopper_pair(n, seed, precision, max_steps=1000, layout=0):
    # planar_layout 0 (default): the lane-group kernel with a group of one lane (mj_planar_lg.hip.h, KL = 1);
    # 1: the one-env-per-lane 9-dof kernel with the ghost leg (mj_cheetah.hip.h)
    pool = DevicePool("Hopper", n, seed=seed, max_episode_steps=max_steps,
                      params={"precision": precision, "planar_layout": layout})
    orc = Oracle("Hopper", n, seed=seed, max_episode_steps=max_steps)
    return pool, orc


def _deep_self_penetration(orc, n, depth=-0.03):
    """Envs whose last forward evaluation holds a body-body contact deeper than `depth`.
    The random folded states below teleport the leg INTO the torso; once two capsule
    axes (which all lie in the y = 0 plane) cross, the segment distance is 0, the
    contact normal is numerical noise (MuJoCo falls back to +x below 1e-15) and the
    outcome is arbitrary in the reference as well.  Real trajectories never get there:
    the contact engages at the 1 mm margin."""
    import ctypes

    from mj_util import _H

    L = orc.lib
    L.mjcpu_raw_contacts.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    inner = ctypes.cast(orc.h, ctypes.POINTER(_H)).contents.h
    out, deep = np.zeros(320), np.zeros(n, bool)
    for e in range(n):
        k = L.mjcpu_raw_contacts(inner, e, out.ctypes.data)
        c = out[:10 * k].reshape(k, 10)
        deep[e] = bool(((c[:, 0] >= 1) & (c[:, 2] < depth)).any())
    return deep


@pytest.mark.parametrize("layout", [0, 1])
def test_hopper_matches_oracle(layout):
    """Reset bit-exact (uniform draws); teacher-forced env-steps (4 RK4 mj_steps) incl.
    folded configurations where the torso-leg / torso-foot / thigh-foot capsule pairs
    touch: obs rtol 1e-9 / atol 1e-10, bookkeeping exact.  Both kernels: the lane-group form with a
    group of one lane (default since round 4) and the 9-dof ghost-leg form."""
    n = 256
    pool, orc = make_hopper_pair(n, 5, 1, layout=layout)
    a, b = hip_reset(pool), orc.reset()
    assert list(a.keys()) == list(b.keys()) and a["obs"].shape == (n, 11)
    np.testing.assert_array_equal(a["obs"], b["obs"])
    # a second oracle without self collisions tells which steps exercised the pair contacts
    ref_noself = Oracle("Hopper", n, seed=5, max_episode_steps=1000,
                        extra=(4, 1e-3, 1, 5e-3, 0, 0, 0, 0, -1, 0, 0, 3, 0, 0, 0, -1, 0, 1, 1))
    ref_noself.reset()
    rng = np.random.default_rng(3)
    worst, seen_term, self_hits, compared = 0.0, False, 0, 0
    for t in range(100):
        st = orc.get_state()
        if t % 4 == 3:  # fold the leg: thigh / knee deep into their range, random height
            st[:, 1] = rng.uniform(0.9, 1.4, n)
            st[:, 2] = rng.uniform(-0.5, 0.5, n)
            st[:, 3] = rng.uniform(-2.6, 0, n)
            st[:, 4] = rng.uniform(-2.6, 0, n)
            st[:, 5] = rng.uniform(-0.78, 0.78, n)
            st[:, 6:12] = rng.uniform(-2, 2, (n, 6))
            st[:, 12:18] = 0
            st[:, 21] = 0  # not done
            orc.set_state(st)
        pool.set_state(st)
        ref_noself.set_state(st)
        act = rng.uniform(-1.2, 1.2, size=(n, 3))
        a, b, c = hip_step(pool, act), orc.step(act), ref_noself.step(act)
        ok = ~_deep_self_penetration(orc, n)
        compared += int(ok.sum())
        np.testing.assert_allclose(a["obs"][ok], b["obs"][ok], rtol=1e-9, atol=1e-10,
                                   err_msg=f"step {t}")
        np.testing.assert_allclose(a["reward"].ravel()[ok], b["reward"].ravel()[ok],
                                   rtol=1e-6, atol=1e-6)
        for k in ("info:x_position", "info:x_velocity"):
            np.testing.assert_allclose(a[k].ravel()[ok], b[k].ravel()[ok], rtol=1e-9, atol=2e-9)
        for k in ("done", "trunc", "elapsed_step", "step_type", "discount"):
            np.testing.assert_array_equal(a[k].ravel()[ok], b[k].ravel()[ok], err_msg=f"{k}@{t}")
        seen_term |= bool((b["done"] & ~b["trunc"]).any())
        self_hits += int((np.abs(b["obs"] - c["obs"]).max(axis=1)[ok] > 1e-9).sum())
        worst = max(worst, float(np.abs(a["obs"] - b["obs"])[ok].max()))
    print(f"Hopper fp64 layout {layout}: worst teacher-forced |d obs| = {worst:.3e}; env-steps with active "
          f"body-body contacts: {self_hits}")
    assert seen_term and self_hits > 100 and compared > 0.8 * n * 100


@pytest.mark.parametrize("layout", [0, 1])
def test_hopper_determinism(layout):
    outs = []
    for _ in range(2):
        p2 = DevicePool("Hopper", 512, seed=3, max_episode_steps=1000, params={"planar_layout": layout})
        hip_reset(p2)
        r2 = np.random.default_rng(5)
        for t in range(30):
            a = hip_step(p2, r2.uniform(-1, 1, size=(512, 3)))
        outs.append(a["obs"].copy())
    np.testing.assert_array_equal(outs[0], outs[1])


def test_hopper_layouts_share_state_and_agree():
    """The two Hopper kernels work on the same device state (qpos / qvel / warm start [9][N], the ghost dofs
    zero): free-running from the same seed they stay within rounding of each other for a short horizon, reset
    rows are bit-identical, and a partly filled last wave (n = 200) is handled."""
    n = 200
    a_pool = DevicePool("Hopper", n, seed=8, max_episode_steps=40, params={"planar_layout": 0})
    b_pool = DevicePool("Hopper", n, seed=8, max_episode_steps=40, params={"planar_layout": 1})
    a, b = hip_reset(a_pool), hip_reset(b_pool)
    rng = np.random.default_rng(1)
    for t in range(12):
        for k in a:
            if t == 0:
                assert a[k].tobytes() == b[k].tobytes(), k
        np.testing.assert_allclose(a["obs"], b["obs"], rtol=1e-7, atol=1e-8, err_msg=f"step {t}")
        for k in ("done", "trunc", "elapsed_step", "step_type"):
            np.testing.assert_array_equal(a[k], b[k], err_msg=f"{k}@{t}")
        act = rng.uniform(-1, 1, size=(n, 3))
        a, b = hip_step(a_pool, act), hip_step(b_pool, act)
    with pytest.raises(Exception, match="planar_layout"):
        DevicePool("Hopper", 8, seed=0, max_episode_steps=10, params={"planar_layout": 2})


# ---- generic observation frame stack (engine-level TypedFrameStackBuffer) ----
@pytest.mark.parametrize("task,adim,amax,S", [("InvertedPendulum", 1, 3.0, 3), ("Swimmer", 2, 1.0, 2),
                                              ("Reacher", 2, 1.0, 4), ("Hopper", 3, 1.0, 3),
                                              ("Walker2d", 6, 1.0, 2)])
def test_chain_families_frame_stack(task, adim, amax, S):
    """frame_stack = S: obs[:, -1] is the newest frame, older frames shift towards 0,
    a reset fills all S slots (envpool/mujoco/frame_stack.h:109-135).  Checked against
    the oracle's un-stacked observations stacked in numpy, across auto-resets.  Hopper / Walker2d: the
    planar pools use the same engine-level ring since round 4 (the lane-group kernels write one frame per
    row); their unhealthy terminations put the auto-reset rows at a different step for every env."""
    n, max_steps = 256, 12
    pool = DevicePool(task, n, seed=2, max_episode_steps=max_steps, params={"frame_stack": S})
    orc = Oracle(task, n, seed=2, max_episode_steps=max_steps)
    a, b = hip_reset(pool), orc.reset()
    nobs = b["obs"].shape[1]
    assert a["obs"].shape == (n, S, nobs)
    ring = np.repeat(b["obs"][:, None, :], S, axis=1)
    np.testing.assert_allclose(a["obs"], ring, rtol=1e-12, atol=1e-14)
    rng = np.random.default_rng(4)
    for t in range(40):
        pool.set_state(orc.get_state())
        act = rng.uniform(-amax, amax, size=(n, adim))
        a, b = hip_step(pool, act), orc.step(act)
        first = b["elapsed_step"].ravel() == 0
        ring = np.concatenate([ring[:, 1:], b["obs"][:, None, :]], axis=1)
        ring[first] = b["obs"][first][:, None, :]
        np.testing.assert_allclose(a["obs"], ring, rtol=1e-9, atol=1e-10, err_msg=f"step {t}")
    # partial / permuted env_id batches keep per-env rings
    ids = rng.permutation(n)[:37].astype(np.int32)
    st = orc.get_state()
    pool.set_state(st)
    act = rng.uniform(-amax, amax, size=(37, adim))
    pool.send(ids, act)
    a = pool.recv_dict()
    full = np.zeros((n, adim))
    full[ids] = act
    b = orc.step(full)
    first = b["elapsed_step"].ravel() == 0
    ring = np.concatenate([ring[:, 1:], b["obs"][:, None, :]], axis=1)
    ring[first] = b["obs"][first][:, None, :]
    np.testing.assert_allclose(a["obs"], ring[ids], rtol=1e-9, atol=1e-10)


# ---- Humanoid / HumanoidStandup (mj_tree.hip.h: HBM workspace, PGS, self collisions) ----
# name, native family, native params, oracle extras (see oracle/mjcpu/tasks.c), obs dim
_HUM_VARIANTS = [
    ("Humanoid-v4", "Humanoid", {"post_constraint": 0}, {}, 376),
    ("Humanoid-v5", "Humanoid",
     {"post_constraint": 1, "use_contact_force": 1, "legacy_healthy_reward": 0,
      "exclude_worldbody_observations": 1, "exclude_root_actuator_forces": 1},
     {12: 1, 13: 1, 14: 1, 15: 0, 19: 1}, 348),
    ("HumanoidStandup-v4", "HumanoidStandup", {"post_constraint": 0}, {}, 376),
    ("HumanoidStandup-v5", "HumanoidStandup",
     {"post_constraint": 1, "exclude_worldbody_observations": 1, "exclude_root_actuator_forces": 1},
     {13: 1, 14: 1, 19: 1}, 348),
]


def _hum_extra(task, over):
    e = [5, 0.1, 1.25 if task == "Humanoid" else 1.0, 0.01, 0, 0, 0, 0, -1, 0, 0, 3, 0, 0, 0, -1,
         0, 1, 0, 0]
    for k, v in over.items():
        e[k] = v
    return e


@pytest.mark.parametrize("name,task,params,over,nobs", _HUM_VARIANTS)
def test_humanoid_matches_oracle(name, task, params, over, nobs):
    """Reset bit-exact (uniform draws only, then one mj_forward: cinert / cvel to rounding);
    teacher-forced env-steps (5 RK4 mj_steps = 20 forward passes with PGS).  PGS is not run to
    convergence, its stopping / revert tests are discontinuous, so two correct implementations
    whose roundings differ can occasionally take one sweep more or less: the bulk must agree
    to rtol 1e-9 / atol 1e-10 and every env-step to 1e-5 relative."""
    n = 192  # three waves
    steps = 60 if task == "Humanoid" else 40
    pool = DevicePool(task, n, seed=5, max_episode_steps=1000, params=params)
    orc = Oracle(task, n, seed=5, max_episode_steps=1000, extra=_hum_extra(task, over))
    a, b = hip_reset(pool), orc.reset()
    assert list(a.keys()) == list(b.keys()) and a["obs"].shape == (n, nobs)
    # obs[1:5] is the root quaternion, normalised in place by mj_kinematics (rounding)
    np.testing.assert_array_equal(a["obs"][:, 0], b["obs"][:, 0])
    np.testing.assert_array_equal(a["obs"][:, 5:45], b["obs"][:, 5:45])
    np.testing.assert_allclose(a["obs"], b["obs"], rtol=1e-11, atol=1e-12)
    for k in a:  # reset rows: zeros, except HumanoidStandup's constant reward_alive
        if k != "obs":
            np.testing.assert_array_equal(a[k].ravel(), b[k].ravel(), err_msg=k)
    rng = np.random.default_rng(3)
    rel_all, seen_term, contact_obs = [], False, 0.0
    for t in range(steps):
        pool.set_state(orc.get_state())
        act = rng.uniform(-0.45, 0.45, size=(n, 17))
        a, b = hip_step(pool, act), orc.step(act)
        rel = (np.abs(a["obs"] - b["obs"]) / (1.0 + np.abs(b["obs"]))).max(axis=1)
        rel_all.append(rel)
        assert rel.max() < 1e-5, (name, t, rel.max())
        np.testing.assert_allclose(a["reward"].ravel(), b["reward"].ravel(), rtol=1e-5, atol=1e-4)
        for k in a:
            if k.startswith("info:") and k not in ("info:env_id", "info:players.env_id"):
                np.testing.assert_allclose(a[k].ravel(), b[k].ravel(), rtol=1e-5, atol=1e-5,
                                           err_msg=f"{k}@{t}")
        for k in ("done", "trunc", "elapsed_step", "step_type", "discount"):
            np.testing.assert_array_equal(a[k].ravel(), b[k].ravel(), err_msg=f"{k}@{t}")
        seen_term |= bool((b["done"] & ~b["trunc"]).any())
        contact_obs = max(contact_obs, float(np.abs(b["obs"][:, -78:]).max()))
    rel_all = np.concatenate(rel_all)
    tight = float((rel_all < 1e-9).mean())
    print(f"{name}: teacher-forced rel |d obs| median {np.median(rel_all):.2e} "
          f"max {rel_all.max():.2e}; within 1e-9: {100 * tight:.2f}%")
    assert tight > 0.99
    if task == "Humanoid":
        assert seen_term  # falls below z = 1.0 under random actions, then auto-resets
    if params.get("post_constraint"):
        assert contact_obs > 1.0  # cfrc_ext is exercised
    else:
        assert contact_obs == 0.0


def test_humanoid_deterministic_and_partial_batches():
    """Same seed, same actions, same batches => bit-identical observations across pools; envs
    stepped through permuted partial batches get the same results as in a full batch (the
    workspace block belongs to the launch's wave, the persistent state to the env) up to
    rounding: which PGS formulation runs is decided per wave."""
    n = 128
    rng = np.random.default_rng(7)
    acts = rng.uniform(-0.4, 0.4, size=(12, n, 17))
    outs = []
    for _ in range(2):
        p = DevicePool("Humanoid", n, seed=11, max_episode_steps=1000, params={"post_constraint": 1})
        hip_reset(p)
        for t in range(12):
            a = hip_step(p, acts[t])
        outs.append(a["obs"].copy())
    np.testing.assert_array_equal(outs[0], outs[1])
    full = DevicePool("Humanoid", n, seed=11, max_episode_steps=1000, params={"post_constraint": 1})
    part = DevicePool("Humanoid", n, seed=11, max_episode_steps=1000, params={"post_constraint": 1})
    hip_reset(full), hip_reset(part)
    for t in range(6):
        a = hip_step(full, acts[t])
        perm = rng.permutation(n).astype(np.int32)
        rows = {}
        for ids in (perm[:37], perm[37:101], perm[101:]):
            part.send(ids, acts[t][ids])
            r = part.recv_dict()
            for j, e in enumerate(r["info:env_id"].ravel()):
                rows[int(e)] = r["obs"][j]
        got = np.stack([rows[e] for e in range(n)])
        np.testing.assert_allclose(got, a["obs"], rtol=1e-9, atol=1e-10, err_msg=f"step {t}")


ALT_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "envpool_amd", "lib",
                       "libenvpool_amd_alt.so")
_ALT_LEG = """
import sys, numpy as np
from envpool_amd.core.device_pool import DevicePool
task, n, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
acts = np.random.default_rng(5).uniform(-0.4, 0.4, size=(25, n, 17))
p = DevicePool(task, n, seed=3, max_episode_steps=1000, params={"post_constraint": 1, "hum_layout": 0})
ids = np.arange(n, dtype=np.int32)
p.reset(ids); p.recv_dict()
seq = []
for t in range(25):
    p.send(ids, acts[t]); a = p.recv_dict()
    assert a["info:env_id"].ravel().tolist() == list(range(n))
    seq.append(np.concatenate([a["obs"], a["reward"].reshape(n, 1), a["done"].reshape(n, 1)], axis=1))
np.save(out, np.stack(seq))
"""


@pytest.mark.parametrize("task", ["Humanoid", "HumanoidStandup"])
def test_humanoid_layouts_and_scheduling_agree(task, tmp_path):
    """The one-env-per-lane-quad kernel (mj_hum4.hip.h, default), with and without the cost-sorted
    scheduling of its waves, and the one-env-per-lane kernel (mj_tree.hip.h, hum_layout=0) are three
    schedules of the same arithmetic: free running from the same seed with the same actions they
    stay together to PGS-rounding level (which solver formulation an env gets depends on the envs
    that share its wave), and the rows come back in send order whatever order the waves ran in.
    The one-env-per-lane kernel lives in the alternate library only (make EPA_ALT_KERNELS=1; the product
    library refuses hum_layout = 0): that leg runs in a subprocess that loads it through ENVPOOL_AMD_LIB."""
    n = 512  # 32 waves of 16 envs: the sort has something to reorder
    rng = np.random.default_rng(5)
    acts = rng.uniform(-0.4, 0.4, size=(25, n, 17))
    outs = []
    for params in ({"hum_layout": 1, "hum_sort": 1}, {"hum_layout": 1, "hum_sort": 0}):
        p = DevicePool(task, n, seed=3, max_episode_steps=1000, params={"post_constraint": 1, **params})
        hip_reset(p)
        seq = []
        for t in range(25):
            a = hip_step(p, acts[t])
            assert a["info:env_id"].ravel().tolist() == list(range(n))
            seq.append(np.concatenate([a["obs"], a["reward"].reshape(n, 1), a["done"].reshape(n, 1)], axis=1))
        outs.append(np.stack(seq))
    with pytest.raises(Exception, match="hum_layout"):
        DevicePool(task, 16, seed=3, max_episode_steps=10, params={"hum_layout": 0})
    pairs = [(outs[1], "sorted vs unsorted waves")]
    if os.path.exists(ALT_LIB):
        import subprocess
        import sys

        out = str(tmp_path / "alt.npy")
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        subprocess.run([sys.executable, "-c", _ALT_LEG, task, str(n), out], check=True, cwd=root,
                       env={**os.environ, "ENVPOOL_AMD_LIB": ALT_LIB, "PYTHONPATH": root})
        pairs.append((np.load(out), "quad vs one-env-per-lane layout"))
    else:
        print("libenvpool_amd_alt.so not built: the one-env-per-lane leg is skipped")
    for other, what in pairs:
        rel = np.abs(outs[0] - other) / (1.0 + np.abs(other))
        # a free-running humanoid amplifies rounding differences: bulk tight, tail bounded
        assert np.median(rel.max(axis=2)) < 1e-10, what
        assert (rel.max(axis=2) < 1e-6).mean() > 0.99, what
        print(f"{task}: {what}: median rel {np.median(rel.max(axis=2)):.1e}, max {rel.max():.1e}")


@pytest.mark.parametrize("task,adim", [("HalfCheetah", 6), ("Hopper", 3), ("Pusher", 7)])
def test_planar_spread_is_bit_identical(task, adim):
    """`planar_spread` only changes which lane of which wave computes an env (16 / 32 / 48 envs per wave
    when the batch is between 16 and 64 envs per SIMD): the arithmetic of an env does not depend on
    its neighbours, so every output is bit-identical to the 64-per-wave launch, for whole batches and
    for partial env_id batches (a 17000-env batch of a 20000-env pool runs 32 per wave)."""
    n = 40000 if task == "Pusher" else 20000  # (Pusher spreads from 32 envs per SIMD up: 48 per wave here)
    rng = np.random.default_rng(11)
    acts = rng.uniform(-1, 1, size=(6, n, adim))
    part = np.sort(rng.choice(n, size=n - 3000, replace=False)).astype(np.int32)
    outs = []
    for spread in (1, 0):
        params = {"planar_spread": spread}
        if task == "HalfCheetah":
            params["planar_layout"] = 1  # the one-env-per-lane kernel (the default is the lane-group one)
        p = DevicePool(task, n, seed=5, max_episode_steps=1000, params=params)
        ids = np.arange(n, dtype=np.int32)
        p.reset(ids)
        seq = [p.recv_dict()]
        for t in range(6):
            if t == 3:
                p.send(part, acts[t][part])
            else:
                p.send(ids, acts[t])
            seq.append(p.recv_dict())
        outs.append(seq)
    for a, b in zip(*outs):
        assert a.keys() == b.keys()
        for k in a:
            assert a[k].shape == b[k].shape and a[k].tobytes() == b[k].tobytes(), k


def test_product_library_refuses_debug_switches():
    """`hum_debug` (stages of the Humanoid quad kernel switched off for timing probes) exists in
    the diagnostic build only: the product library must not produce a figure with physics disabled."""
    with pytest.raises(Exception, match="hum_debug"):
        DevicePool("Humanoid", 64, seed=0, max_episode_steps=1000, params={"hum_debug": 1})


# ---- one env per LANE GROUP (mj_planar_lg.hip.h): HalfCheetah / Walker2d, fp64 --------------------
@pytest.mark.parametrize("layout,waves", [(2, 1), (4, 2), (4, 1)])
@pytest.mark.parametrize("task,otask", [("HalfCheetah", "HalfCheetah"), ("Walker2d", "Walker2d"),
                                        ("Walker2d", "Walker2dV5")])
def test_planar_lane_group_matches_oracle(task, otask, layout, waves):
    """Every variant of the lane-group kernel (2 or 4 lanes per env; at 4 lanes the register budget for 1 or
    2 waves per SIMD -- at 2 lanes LDS allows one wave whatever the budget, so (2, 2) is the (2, 1) code and is
    not a case), teacher forced against the oracle: obs rtol 1e-9 / atol 1e-10, info keys, bookkeeping exact.
    n = 200 leaves the last wave partially filled; resets are bit-identical to the one-env-per-lane
    kernel's (the group's first lane makes the same mt19937 draws)."""
    n, steps = 200, 60
    params = {"precision": 1, "planar_layout": layout, "planar_waves": waves}
    if task == "Walker2d":
        params.update(xml_v5=1 if otask == "Walker2dV5" else 0,
                      legacy_healthy_reward=0 if otask == "Walker2dV5" else 1)
    pool = DevicePool(task, n, seed=9, max_episode_steps=1000, params=params)
    lane = DevicePool(task, n, seed=9, max_episode_steps=1000, params={**params, "planar_layout": 1})
    orc = Oracle(otask, n, seed=9, max_episode_steps=1000)
    a, l, b = hip_reset(pool), hip_reset(lane), orc.reset()
    for k in a:
        assert a[k].tobytes() == l[k].tobytes(), k
    rng = np.random.default_rng(3)
    worst, worst_lane = 0.0, 0.0
    info = ("info:x_position", "info:x_velocity") + (("info:reward_ctrl", "info:reward_run") if task == "HalfCheetah" else ())
    for t in range(steps):
        st = orc.get_state()
        pool.set_state(st), lane.set_state(st)
        act = rng.uniform(-1.2, 1.2, size=(n, 6))
        a, l, b = hip_step(pool, act), hip_step(lane, act), orc.step(act)
        np.testing.assert_allclose(a["obs"], b["obs"], rtol=1e-9, atol=1e-10, err_msg=f"step {t}")
        np.testing.assert_allclose(a["reward"].ravel(), b["reward"].ravel(), rtol=1e-6, atol=1e-6)
        for k in info:
            np.testing.assert_allclose(a[k].ravel(), b[k].ravel(), rtol=1e-9, atol=2e-9, err_msg=k)
        for k in ("done", "trunc", "elapsed_step", "step_type", "discount", "info:env_id"):
            np.testing.assert_array_equal(a[k].ravel(), b[k].ravel(), err_msg=f"{k}@{t}")
        worst = max(worst, float(np.abs(a["obs"] - b["obs"]).max()))
        worst_lane = max(worst_lane, float(np.abs(a["obs"] - l["obs"]).max()))
    print(f"{otask} layout {layout} waves {waves}: worst |d obs| vs oracle {worst:.2e}, vs the "
          f"one-env-per-lane kernel {worst_lane:.2e}")


@pytest.mark.parametrize("task", ["HalfCheetah", "Walker2d"])
@pytest.mark.parametrize("layout", [2, 4])
def test_planar_lane_group_batch_independent(task, layout):
    """An env's arithmetic does not depend on the wave it lands in: whole batches, permuted partial
    env_id batches and a second pool stepping only a subset give bit-identical rows; two runs agree."""
    n = 5000
    rng = np.random.default_rng(2)
    acts = rng.uniform(-1, 1, size=(8, n, 6))
    sub = np.sort(rng.choice(n, size=777, replace=False)).astype(np.int32)
    perm = rng.permutation(sub).astype(np.int32)
    ids = np.arange(n, dtype=np.int32)
    params = {"planar_layout": layout}
    full = [DevicePool(task, n, seed=5, max_episode_steps=6, params=params) for _ in range(2)]
    part = DevicePool(task, n, seed=5, max_episode_steps=6, params=params)
    for p in full + [part]:
        p.reset(ids)
    f0, f1, pp = full[0].recv_dict(), full[1].recv_dict(), part.recv_dict()
    for t in range(8):
        for k in f0:
            assert f0[k].tobytes() == f1[k].tobytes(), (k, t)
        full[0].send(ids, acts[t]), full[1].send(ids, acts[t])
        f0, f1 = full[0].recv_dict(), full[1].recv_dict()
        # the partial pool steps the subset in a permuted order (rows come back in send order),
        # then the rest
        part.send(perm, acts[t][perm])
        a = part.recv_dict()
        for k in f0:
            assert a[k].tobytes() == np.ascontiguousarray(f0[k][perm]).tobytes(), (k, t)
        rest = np.setdiff1d(ids, sub).astype(np.int32)
        part.send(rest, acts[t][rest])
        part.recv_dict()
    assert np.array_equal(full[0].get_state(), part.get_state())


@pytest.mark.parametrize("task,adim,amax,exact", [("HalfCheetah", 6, 1.0, 8), ("Hopper", 3, 1.0, 11), ("Walker2d", 6, 1.0, 17),
                                                   ("Ant", 8, 1.0, 13), ("InvertedPendulum", 1, 3.0, 4),
                                                   ("Humanoid", 17, 0.4, 22)])
def test_reset_draws_across_generator_wraps(task, adim, amax, exact):
    """The per-env mt19937 is regenerated lazily (one word per draw in the [624][N] layout -- HalfCheetah: every env
    draws in the same launch --, one 16-word tile at a time for the families whose envs reset at their own times).
    With max_episode_steps = 1 every env resets every second step; a reset draws 20 - 100 words (uniform positions,
    normal or uniform velocities), so 260 steps walk every env's 624-word block 4 - 20 times.  The first `exact`
    observations of a reset row are uniform draws -- bit-exact against the oracle (libstdc++ std::mt19937 +
    distributions); the rest go through log / sqrt (normal draws: HalfCheetah, Ant) or mj_forward (Humanoid) and agree
    to rounding -- a generator out of step would show as errors of the size of the noise, 1e-2 .. 1e-1."""
    n, steps = 64, 260
    pool = DevicePool(task, n, seed=23, max_episode_steps=1)
    orc = Oracle(task, n, seed=23, max_episode_steps=1)
    a, b = hip_reset(pool), orc.reset()
    rng = np.random.default_rng(9)
    resets = 0
    for t in range(steps):
        rows = np.nonzero(b["elapsed_step"].ravel() == 0)[0]
        resets += len(rows)
        assert np.array_equal(a["elapsed_step"].ravel(), b["elapsed_step"].ravel()), (task, t)
        ao, bo = np.asarray(a["obs"])[rows], np.asarray(b["obs"])[rows]
        assert np.array_equal(np.ascontiguousarray(ao[:, :exact]).view(np.uint8),
                              np.ascontiguousarray(bo[:, :exact]).view(np.uint8)), (task, t)
        np.testing.assert_allclose(ao, bo, rtol=1e-9, atol=1e-11, err_msg=f"{task} t={t}")
        act = rng.uniform(-amax, amax, size=(n, adim))
        a, b = hip_step(pool, act), orc.step(act)
    assert resets >= n * (steps // 2)


@pytest.mark.parametrize("task", ["HalfCheetah", "Walker2d", "Hopper", "Ant"])
def test_one_evaluation_solvers_on_adversarial_states(task):
    """the one-evaluation line search (mj_planar_lg.hip.h / mj_ant4.hip.h: one Newton step of the 1-D problem
    per trip, unverified; exact search only as a fallback from trip 8) ON THE DEVICE, far from the benchmark's states
    -- deep penetrations, joints beyond their ranges, velocities ~ N(0, 8), a garbage warm start -- against the oracle,
    whose solver searches its lines exactly: one env-step from each state, same result to 1e-9 (the CPU twins:
    tests/test_mjcpu_invariants.py::test_lane_group_solver_on_adversarial_states / test_ant_solver_...)."""
    from oracle.orc import Oracle

    n = 512
    rng = np.random.default_rng(21)
    orc = Oracle(task, n, seed=5, max_episode_steps=1000)
    orc.reset()
    pool = DevicePool(task, n, seed=5, max_episode_steps=1000)
    ids = np.arange(n, dtype=np.int32)
    pool.reset(ids)
    pool.recv()
    for rep in range(3):
        st = orc.get_state()
        if task == "Ant":
            nq, nv, adim = 15, 14, 8
            st[:, 2] = rng.uniform(0.15, 0.9, n)
            quat = rng.normal(0, 1, (n, 4))
            st[:, 3:7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
            st[:, 7:15] = rng.uniform(-1.2, 1.2, (n, 8))
            st[:, nq:nq + nv] = rng.normal(0, 4, (n, nv))
            st[:, nq + nv:nq + 2 * nv] = rng.normal(0, 30, (n, nv))
        else:
            nd = 6 if task == "Hopper" else 9
            adim = nd - 3
            st[:, 1] = rng.uniform(-0.3, 0.6, n) if task == "HalfCheetah" else rng.uniform(0.3, 1.5, n)
            st[:, 2] = rng.uniform(-3, 3, n)
            st[:, 3:nd] = rng.uniform(-1.5, 1.5, (n, nd - 3))
            st[:, nd:2 * nd] = rng.normal(0, 8, (n, nd))
            st[:, 2 * nd:3 * nd] = rng.normal(0, 50, (n, nd))
        orc.set_state(st)
        pool.set_state(orc.get_state())
        act = rng.uniform(-1, 1, (n, adim))
        b = orc.step(act)
        pool.send(ids, act)
        a = pool.recv_dict()
        ok = (b["elapsed_step"].ravel() > 0) & np.isfinite(b["obs"]).all(axis=1)
        assert ok.sum() > n // 2
        ref, got = b["obs"][ok], a["obs"][ok]
        err = np.abs(got - ref) / (1 + np.abs(ref))
        assert err.max() < 1e-9, (task, rep, float(err.max()))
        orc.reset(), pool.reset(ids), pool.recv()
