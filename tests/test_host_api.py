"""CPU tests of the host side: registry, spec export, adaptors, C-ABI exports.

The adaptors are exercised against a fake `DevicePool` (no compute is done on
the CPU by the product: `make()` without a GPU must fail loudly, which is also
asserted here).  Modelled on the reference's API-contract tests
(envpool/atari/api_test.py:120-330, classic_control_test.py:85-121,
make_test.py).
"""
import ctypes
import re

import numpy as np
import pytest

import envpool_amd as envpool
from envpool_amd.core import binding, native


def test_list_all_envs_and_registry():
    ids = envpool.list_all_envs()
    for t in ["CartPole-v0", "CartPole-v1", "Pendulum-v0", "Pendulum-v1",
              "MountainCar-v0", "MountainCarContinuous-v0", "Acrobot-v1", "Catch-v0",
              "FrozenLake-v1", "FrozenLake8x8-v1", "Taxi-v3", "NChain-v0",
              "CliffWalking-v0", "CliffWalking-v1", "CliffWalkingSlippery-v1",
              "Blackjack-v1", "HalfCheetah-v3", "HalfCheetah-v4", "HalfCheetah-v5",
              "Ant-v3", "Ant-v4", "Ant-v5", "Walker2d-v3", "Walker2d-v4", "Walker2d-v5",
              "Hopper-v3", "Hopper-v4", "Hopper-v5", "Swimmer-v3", "Swimmer-v4", "Swimmer-v5",
              "Reacher-v2", "Reacher-v4", "Reacher-v5", "Pusher-v2", "Pusher-v4", "Pusher-v5",
              "InvertedPendulum-v2",
              "InvertedPendulum-v4", "InvertedPendulum-v5", "InvertedDoublePendulum-v2",
              "InvertedDoublePendulum-v4", "InvertedDoublePendulum-v5",
              "Humanoid-v3", "Humanoid-v4", "Humanoid-v5", "HumanoidStandup-v2",
              "HumanoidStandup-v4", "HumanoidStandup-v5"]:
        assert t in ids, t
    with pytest.raises(AssertionError):
        envpool.make("NoSuchEnv-v0", "gym", num_envs=1)
    with pytest.raises(AssertionError):
        envpool.make("CartPole-v1", "foo", num_envs=1)


def test_spec_config_defaults_and_key_order():
    spec = envpool.make_spec("CartPole-v1", num_envs=8)
    c = spec.config
    # common_config prefix, envpool/core/env_spec.h:26-31
    assert list(c._fields[:10]) == [
        "num_envs", "batch_size", "num_threads", "max_num_players",
        "thread_affinity_offset", "base_path", "seed", "env_seed",
        "gym_reset_return_info", "max_episode_steps"]
    assert c.num_envs == 8 and c.batch_size == 8  # batch_size 0 -> num_envs
    assert c.seed == 42 and c.max_episode_steps == 500 and c.reward_threshold == 475.0
    assert spec._action_keys == ["env_id", "players.env_id", "action"]
    assert spec._state_keys[:8] == [
        "info:env_id", "info:players.env_id", "elapsed_step", "done", "reward",
        "discount", "step_type", "trunc"]
    assert spec._state_keys[8:] == ["obs"]
    assert envpool.make_spec("Acrobot-v1")._state_keys[8:] == ["obs", "info:state"]
    assert envpool.make_spec("HalfCheetah-v4")._state_keys[8:] == [
        "obs", "info:reward_run", "info:reward_ctrl", "info:x_position",
        "info:x_velocity"]
    hc = envpool.make_spec("HalfCheetah-v4").config
    assert (hc.frame_skip, hc.post_constraint, hc.max_episode_steps) == (5, False, 1000)
    assert envpool.make_spec("HalfCheetah-v5").config.post_constraint is True
    # version-specific registrations (envpool/mujoco/gym/registration.py:36-83)
    a3, a5 = envpool.make_spec("Ant-v3").config, envpool.make_spec("Ant-v5").config
    assert (a3.use_contact_force, a3.post_constraint) == (True, False)
    assert (a5.use_contact_force, a5.exclude_worldbody_contact_forces,
            a5.legacy_healthy_reward, a5.post_constraint) == (True, True, False, True)
    assert envpool.make_spec("Ant-v3").observation_space.shape == (111,)
    assert envpool.make_spec("Ant-v5").observation_space.shape == (105,)
    w5 = envpool.make_spec("Walker2d-v5").config
    assert (w5.xml_file, w5.legacy_healthy_reward) == ("walker2d_v5.xml", False)
    assert envpool.make_spec("Hopper-v5").config.legacy_healthy_reward is False
    r5 = envpool.make_spec("Reacher-v5")
    assert r5.config.reward_after_step is True and r5.observation_space.shape == (10,)
    assert envpool.make_spec("Reacher-v4").config.max_episode_steps == 50
    p5 = envpool.make_spec("Pusher-v5")
    assert (p5.config.xml_file, p5.config.reward_after_step, p5.config.weighted_reward_info,
            p5.config.max_episode_steps) == ("pusher_v5.xml", True, True, 100)
    assert p5.observation_space.shape == (23,) and p5.action_space.shape == (7,)
    assert envpool.make_spec("Pusher-v4").config.xml_file == "pusher.xml"
    d5 = envpool.make_spec("InvertedDoublePendulum-v5")
    assert d5.config.constraint_obs_dim == 1 and d5.observation_space.shape == (9,)
    assert envpool.make_spec("InvertedPendulum-v5").config.reward_if_not_terminated is True
    # humanoid.h:50-60: 376 observations, v5 drops the world body rows and the root actuator forces
    h4, h5 = envpool.make_spec("Humanoid-v4"), envpool.make_spec("Humanoid-v5")
    assert h4.observation_space.shape == (376,) and h5.observation_space.shape == (348,)
    assert (h4.config.post_constraint, h5.config.post_constraint) == (False, True)
    assert envpool.make_spec("Humanoid-v3").config.use_contact_force is True
    assert h4.action_space.shape == (17,) and float(h4.action_space.high[0]) == 0.4
    assert envpool.make_spec("HumanoidStandup-v5").observation_space.shape == (348,)
    assert envpool.make_spec("HumanoidStandup-v4")._state_keys[8:] == [
        "obs", "info:reward_linup", "info:reward_quadctrl", "info:reward_alive",
        "info:reward_impact"]
    assert envpool.make_spec("Swimmer-v4", frame_stack=3).observation_space.shape == (3, 8)
    with pytest.raises(ValueError):
        envpool.make_spec("Walker2d-v4", xml_file="walker2d_custom.xml")


def test_spaces_and_dm_specs():
    s = envpool.make_spec("CartPole-v1")
    assert s.observation_space.shape == (4,) and s.observation_space.dtype == np.float32
    assert s.action_space.n == 2
    assert np.isclose(s.observation_space.high[0], 4.8)
    assert envpool.make_spec("Pendulum-v1").action_space.shape == (1,)
    assert envpool.make_spec("FrozenLake8x8-v1").observation_space.n == 64
    assert envpool.make_spec("Taxi-v3").observation_space.n == 500
    assert envpool.make_spec("Catch-v0").observation_space.shape == (10, 5)
    hc = envpool.make_spec("HalfCheetah-v4")
    assert hc.observation_space.shape == (17,) and hc.observation_space.dtype == np.float64
    assert hc.action_space.shape == (6,) and hc.action_space.low.min() == -1.0
    obs_spec = s.observation_spec()
    assert obs_spec._fields == ("env_id", "players", "obs")
    assert s.action_spec().num_values == 2
    assert s.reward_threshold == 475.0
    assert envpool.make_spec("NChain-v0").reward_threshold is None


def test_kwargs_validation():
    with pytest.raises(AssertionError):
        envpool.make_spec("CartPole-v1", num_envs=0)
    with pytest.raises(AssertionError):
        envpool.make_spec("CartPole-v1", num_envs=4, batch_size=5)
    with pytest.raises(TypeError):
        envpool.make_spec("CartPole-v1", no_such_key=1)
    with pytest.raises(AssertionError):
        envpool.make_spec("CartPole-v1", num_envs=2, seed=[1, 2, 3])
    sp = envpool.make_spec("CartPole-v1", num_envs=3, seed=[5, 6, 7])
    assert sp.config.env_seed == [5, 6, 7] and sp.config.seed == 0
    with pytest.raises(ValueError):
        envpool.make("CartPole-v1", "gym", num_envs=1, gym_reset_return_info=False)
    with pytest.raises(ValueError):
        envpool.make("CartPole-v1", "gym", num_envs=1, render_mode="ansi")
    with pytest.raises(ValueError):
        envpool.make_spec("HalfCheetah-v4", frame_stack=0)
    with pytest.raises(ValueError):
        envpool.make_spec("HalfCheetah-v4", xml_file="other.xml")
    assert envpool.make_spec("HalfCheetah-v4", frame_stack=4).observation_space.shape == (4, 17)


def test_no_cpu_fallback():
    """Without a GPU the product must refuse to run (no silent CPU path)."""
    if native.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        envpool.make("CartPole-v1", "gym", num_envs=2)


def test_c_abi_exports_every_declared_symbol():
    header = open("include/envpool_amd.h").read()
    declared = set(re.findall(r"\b(epa_[a-z_0-9]+)\s*\(", header))
    lib = ctypes.CDLL(native.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/envpool_amd.h but not exported"
    assert declared == set(native.EXPORTED_SYMBOLS)


FAMILY_PARAMS = {
    "CartPole-v1": {}, "Pendulum-v1": {"version": 1}, "MountainCar-v0": {},
    "MountainCarContinuous-v0": {}, "Acrobot-v1": {}, "Catch-v0": {"height": 10, "width": 5},
    "FrozenLake8x8-v1": {"size": 8}, "Taxi-v3": {}, "NChain-v0": {},
    "CliffWalkingSlippery-v1": {"is_slippery": 1}, "Blackjack-v1": {}, "HalfCheetah-v4": {},
    "Ant-v4": {},
    "Ant-v3": {"use_contact_force": 1},
    "Ant-v5": {"use_contact_force": 1, "exclude_worldbody_contact_forces": 1, "post_constraint": 1},
    "Walker2d-v4": {}, "Walker2d-v5": {"xml_v5": 1}, "Hopper-v4": {}, "Swimmer-v4": {},
    "Reacher-v4": {}, "Reacher-v5": {"obs_include_z_distance": 0},
    "Pusher-v4": {}, "Pusher-v5": {"xml_v5": 1, "reward_after_step": 1, "weighted_reward_info": 1},
    "InvertedPendulum-v4": {}, "InvertedDoublePendulum-v4": {},
    "InvertedDoublePendulum-v5": {"constraint_obs_dim": 1},
    "Humanoid-v4": {}, "HumanoidStandup-v4": {},
    "Humanoid-v5": {"exclude_worldbody_observations": 1, "exclude_root_actuator_forces": 1},
}
NATIVE = {"CartPole-v1": "CartPole", "Pendulum-v1": "Pendulum", "MountainCar-v0": "MountainCar",
          "MountainCarContinuous-v0": "MountainCarContinuous", "Acrobot-v1": "Acrobot",
          "Catch-v0": "Catch", "FrozenLake8x8-v1": "FrozenLake", "Taxi-v3": "Taxi",
          "NChain-v0": "NChain", "CliffWalkingSlippery-v1": "CliffWalking",
          "Blackjack-v1": "Blackjack", "HalfCheetah-v4": "HalfCheetah", "Ant-v4": "Ant",
          "Ant-v3": "Ant", "Ant-v5": "Ant", "Walker2d-v4": "Walker2d", "Walker2d-v5": "Walker2d",
          "Hopper-v4": "Hopper", "Swimmer-v4": "Swimmer", "Reacher-v4": "Reacher",
          "Reacher-v5": "Reacher", "Pusher-v4": "Pusher", "Pusher-v5": "Pusher",
          "InvertedPendulum-v4": "InvertedPendulum",
          "InvertedDoublePendulum-v4": "InvertedDoublePendulum",
          "InvertedDoublePendulum-v5": "InvertedDoublePendulum",
          "Humanoid-v4": "Humanoid", "Humanoid-v5": "Humanoid",
          "HumanoidStandup-v4": "HumanoidStandup"}


@pytest.mark.parametrize("task", sorted(FAMILY_PARAMS))
def test_python_spec_matches_c_abi_layout(task):
    """Key order, dtypes and row shapes exported by epa_describe_* must equal
    the Python spec tables (the reference's -1 player dim dropped)."""
    spec = envpool.make_spec(task)
    for which, keys, specs in (("state", spec._state_keys, spec._state_spec),
                               ("action", spec._action_keys, spec._action_spec)):
        got = native.describe(NATIVE[task], FAMILY_PARAMS[task], which)
        assert [g[0] for g in got] == list(keys)
        for (name, dtype, shape), s in zip(got, specs):
            assert np.dtype(dtype) == s[0], name
            assert list(shape) == [d for d in s[1] if d != -1], name


class FakeDevicePool:
    """Deterministic stand-in for DevicePool: echoes ids, counts steps."""

    def __init__(self, family, num_envs, batch_size=0, seed=42, env_seed=None,
                 max_episode_steps=0, device=0, env_id_offset=0, params=None):
        self.num_envs = num_envs
        self.state_keys = native.describe(family, params, "state")
        self.pending = []
        self.t = np.zeros(num_envs, dtype=np.int32)
        self.sent = []

    def _make(self, ids, reset):
        out = []
        for name, dtype, shape in self.state_keys:
            a = np.zeros((len(ids), *shape), dtype=dtype)
            if name in ("info:env_id", "info:players.env_id"):
                a[:] = ids
            elif name == "elapsed_step":
                a[:] = self.t[ids]
            elif name == "done":
                a[:] = self.t[ids] >= 3
            elif name == "trunc":
                a[:] = self.t[ids] >= 3
            elif name == "reward":
                a[:] = 0.0 if reset else 1.0
            elif name == "step_type":
                a[:] = 0 if reset else 1
            out.append(a)
        return out

    def send(self, env_id, action):
        self.sent.append((np.array(env_id), np.array(action)))
        self.t[env_id] += 1
        self.pending.append(self._make(np.asarray(env_id), False))

    def reset(self, ids):
        self.t[ids] = 0
        self.pending.append(self._make(np.asarray(ids), True))

    def recv(self):
        return self.pending.pop(0)

    def close(self):
        pass


@pytest.fixture
def fake_pool(monkeypatch):
    monkeypatch.setattr(binding, "DevicePool", FakeDevicePool)


def test_gymnasium_adaptor_contract(fake_pool):
    env = envpool.make("CartPole-v1", "gym", num_envs=4)
    assert len(env) == 4 and env.num_envs == 4 and not env.is_async
    obs, info = env.reset()
    assert obs.shape == (4, 4) and obs.dtype == np.float32
    assert set(info) == {"env_id", "players", "elapsed_step"}
    assert info["players"]["env_id"].tolist() == [0, 1, 2, 3]
    out = env.step(np.array([0, 1, 0, 1]))  # int64 gets cast: envpool.py:192-197
    obs, rew, term, trunc, info = out
    assert rew.dtype == np.float32 and term.dtype == np.bool_ and trunc.dtype == np.bool_
    assert env._pool.sent[-1][1].dtype == np.int32
    # terminated = done & ~trunc (gymnasium_envpool.py:227)
    for _ in range(2):
        obs, rew, term, trunc, info = env.step(np.zeros(4, dtype=np.int32))
    assert trunc.all() and not term.any()
    # partial env_id stepping keeps send order
    obs, rew, term, trunc, info = env.step(np.zeros(2, dtype=np.int32), np.array([3, 1]))
    assert info["env_id"].tolist() == [3, 1]
    # dict actions with explicit env_id
    env.send({"action": np.zeros(1, dtype=np.int32), "env_id": np.array([2], dtype=np.int32)})
    assert env.recv()[4]["env_id"].tolist() == [2]
    with pytest.raises(RuntimeError):
        env.xla()
    with pytest.raises(RuntimeError):
        env.render()
    with pytest.warns(UserWarning):
        env.reset(seed=3)
    o, i = env.reset(options={"reset_mask": [True, False, True, False]})
    assert i["env_id"].tolist() == [0, 2]
    with pytest.raises(ValueError):
        env.reset(options={"bogus": 1})
    assert "num_envs=4" in repr(env)


def test_action_checks(fake_pool):
    env = envpool.make("Pendulum-v1", "gym", num_envs=3)
    env.reset()
    with pytest.raises(RuntimeError, match="Expected shape"):
        env.step(np.zeros((3, 2), dtype=np.float32))
    env = envpool.make("Pendulum-v1", "gym", num_envs=3)
    env.reset()
    with pytest.raises(RuntimeError, match="Expected dtype"):
        env.send({"action": np.zeros((3, 1), dtype=np.float64)})


def test_dm_adaptor_contract(fake_pool):
    env = envpool.make_dm("Acrobot-v1", num_envs=2)
    ts = env.reset()
    assert ts.step_type.tolist() == [0, 0] and ts.first().all()
    assert ts.observation._fields == ("env_id", "players", "obs", "state")
    assert ts.observation.obs.shape == (2, 6) and ts.observation.state.shape == (2, 2)
    ts = env.step(np.array([0, 2], dtype=np.int32))
    assert ts.reward.dtype == np.float32 and ts.discount.shape == (2,)
    assert env.action_spec().num_values == 3
    assert env.observation_spec().obs.shape == (6,)


def test_default_config_tables_match_reference_headers():
    """tests/golden/spec_defaults.json was extracted from the reference's C++
    `DefaultConfig()` tables (tests/golden/make_spec_golden.py): every family's Python
    table must list the same keys, in the same order, with the same defaults; the only
    additions allowed are this engine's extension keys at the end."""
    import importlib
    import json
    import os

    from envpool_amd.core.binding import FamilyDef

    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "spec_defaults.json")))
    fams = {}
    for mod in ("envpool_amd.classic_control", "envpool_amd.toy_text", "envpool_amd.mujoco.gym"):
        m = importlib.import_module(mod)
        for v in vars(m).values():
            if isinstance(v, FamilyDef):
                fams[v.name] = v
    assert set(gold) <= set(fams), set(gold) - set(fams)
    extensions = {"precision"}
    for name, g in gold.items():
        mine = [(k, v) for k, v in fams[name].default_config]
        ref = [tuple(kv) for kv in g["default_config"]]
        assert mine[:len(ref)] == ref, (name, g["source"], mine[:len(ref)], ref)
        assert {k for k, _ in mine[len(ref):]} <= extensions, (name, mine[len(ref):])


def test_registry_matches_reference_registration_modules():
    """tests/golden/registry.json records what the reference's own registration modules
    pass to `register` (tests/golden/make_registry_golden.py).  Every id this engine
    registers must carry the same class names and keyword arguments (aliases included);
    and every reference id must be registered."""
    import json
    import os

    from envpool_amd.registration import registry

    envpool.list_all_envs()  # imports every registration module
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "registry.json")))
    missing = sorted(set(gold) - set(registry.specs))
    assert missing == [], missing
    checked = 0
    for task_id, g in gold.items():
        if task_id not in registry.specs:
            continue
        import_path, spec_cls, kwargs = registry.specs[task_id]
        assert import_path == g["import_path"].replace("envpool.", "envpool_amd.", 1)
        assert spec_cls == g["spec_cls"]
        assert registry.envpools[task_id]["dm"][1] == g["dm_cls"]
        assert registry.envpools[task_id]["gymnasium"][1] == g["gymnasium_cls"]
        want = dict(g["kwargs"])
        for alias in want.pop("aliases", []):
            assert registry.specs[alias][1] == spec_cls, alias
        mine = {k: v for k, v in kwargs.items() if k != "base_path"}
        assert mine == want, (task_id, mine, want)
        checked += 1
    assert checked >= 40


def test_affinity_helpers_without_a_device():
    """`bind_host_to_device` (host placement on a multi-socket box, the reference's numactl recipe): the cpulist parser,
    and that a box without a device (or without NUMA information) is left alone."""
    import os

    from envpool_amd.core import affinity

    assert affinity._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert affinity._parse_cpulist("") == []
    before = os.sched_getaffinity(0)
    out = envpool.bind_host_to_device(0)
    if not out["bound"]:
        assert os.sched_getaffinity(0) == before and out["cpus"] == 0
    else:  # (a GPU box: bound to a non-empty subset of what it had)
        now = os.sched_getaffinity(0)
        assert now and now <= before
        os.sched_setaffinity(0, before)
