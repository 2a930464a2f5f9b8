"""The reference's own pybind11 host shim over libenvpool_amd.so.

integration/_build/refbind*.so is `envpool/core/py_envpool.h` (PyEnvSpec, PyEnvPool, the
REGISTER macro: py_envpool.h:206-332) compiled IN PLACE from /root/reference, with
`DeviceEnvPool<Spec>` (integration/refbind/device_envpool.h, an `EnvPool<Spec>` subclass,
envpool/core/envpool.h:29-56) where the reference instantiates `AsyncEnvPool<Env>`, and the
reference's own Spec types (CartPoleEnvSpec, PendulumEnvSpec, FrozenLakeEnvSpec,
HalfCheetahEnvSpec, AntEnvSpec).  north_star: "the pybind11 host shim stays".

CPU part (no GPU): the module loads and its spec surface -- produced by the REFERENCE's
EnvFns -- equals what envpool_amd's ctypes binding (core/binding.py) manufactures, i.e. the
Python spec tables are pinned against the real C++ spec objects, not only against the
extracted JSON.  GPU part: same seed + same actions through `_send/_recv/_reset` of the
pybind11 classes and through the ctypes path are bit-identical, the pybind11 classes slot
into the Python adaptors (`py_env`), and the C++ driver that follows
envpool/mujoco/gym/mujoco_gym_envpool_test.cc passes.
"""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "integration", "_build")


def _refbind():
    if not any(f.startswith("refbind") and f.endswith(".so") for f in os.listdir(BUILD)
               ) if os.path.isdir(BUILD) else True:
        pytest.skip("integration/_build/refbind*.so not built (needs /root/reference: "
                    "make -C integration)")
    if BUILD not in sys.path:
        sys.path.insert(0, BUILD)
    return importlib.import_module("refbind")


# refbind class stem, envpool_amd module, native class stem, task id for make()
FAMILIES = [
    ("CartPole", "envpool_amd.classic_control", "CartPole", "CartPole-v1"),
    ("Pendulum", "envpool_amd.classic_control", "Pendulum", "Pendulum-v1"),
    ("MountainCar", "envpool_amd.classic_control", "MountainCar", "MountainCar-v0"),
    ("MountainCarContinuous", "envpool_amd.classic_control", "MountainCarContinuous",
     "MountainCarContinuous-v0"),
    ("Acrobot", "envpool_amd.classic_control", "Acrobot", "Acrobot-v1"),
    ("Catch", "envpool_amd.toy_text", "Catch", "Catch-v0"),
    ("FrozenLake", "envpool_amd.toy_text", "FrozenLake", "FrozenLake-v1"),
    ("Taxi", "envpool_amd.toy_text", "Taxi", "Taxi-v3"),
    ("NChain", "envpool_amd.toy_text", "NChain", "NChain-v0"),
    ("CliffWalking", "envpool_amd.toy_text", "CliffWalking", "CliffWalking-v0"),
    ("Blackjack", "envpool_amd.toy_text", "Blackjack", "Blackjack-v1"),
    ("GymHalfCheetah", "envpool_amd.mujoco.gym", "GymHalfCheetah", "HalfCheetah-v4"),
    ("GymAnt", "envpool_amd.mujoco.gym", "GymAnt", "Ant-v4"),
    ("GymWalker2d", "envpool_amd.mujoco.gym", "GymWalker2d", "Walker2d-v4"),
    ("GymHopper", "envpool_amd.mujoco.gym", "GymHopper", "Hopper-v4"),
    ("GymSwimmer", "envpool_amd.mujoco.gym", "GymSwimmer", "Swimmer-v4"),
    ("GymReacher", "envpool_amd.mujoco.gym", "GymReacher", "Reacher-v4"),
    ("GymPusher", "envpool_amd.mujoco.gym", "GymPusher", "Pusher-v4"),
    ("GymInvertedPendulum", "envpool_amd.mujoco.gym", "GymInvertedPendulum", "InvertedPendulum-v4"),
    ("GymInvertedDoublePendulum", "envpool_amd.mujoco.gym", "GymInvertedDoublePendulum",
     "InvertedDoublePendulum-v4"),
    ("GymHumanoid", "envpool_amd.mujoco.gym", "GymHumanoid", "Humanoid-v4"),
    ("GymHumanoidStandup", "envpool_amd.mujoco.gym", "GymHumanoidStandup", "HumanoidStandup-v4"),
]


def _ours(module, stem):
    """(native spec class, native pool class) of envpool_amd's ctypes binding."""
    mod = importlib.import_module(module)
    spec_cls = getattr(mod, f"{stem}EnvSpec")
    pool_cls = getattr(mod, f"{stem}GymnasiumEnvPool")
    native_spec = [b for b in spec_cls.__mro__ if b.__name__ == f"_{stem}EnvSpec"][0]
    native_pool = [b for b in pool_cls.__mro__ if b.__name__ == f"_{stem}EnvPool"][0]
    return native_spec, native_pool


def _norm(x):
    """spec tuples -> comparable plain python (dtype str, lists, floats)."""
    if isinstance(x, np.dtype):
        return str(x)
    if isinstance(x, (tuple, list)):
        return [_norm(v) for v in x]
    if isinstance(x, (np.floating, float)):
        return float(x)
    if isinstance(x, (np.integer, int)) and not isinstance(x, bool):
        return int(x)
    return x


@pytest.mark.parametrize("stem,module,ours,task", FAMILIES, ids=[f[0] for f in FAMILIES])
def test_spec_surface_equals_the_reference_pybind_classes(stem, module, ours, task):
    rb = _refbind()
    ref_spec = getattr(rb, f"_{stem}EnvSpec")
    ref_pool = getattr(rb, f"_{stem}EnvPool")
    our_spec, our_pool = _ours(module, ours)
    # REGISTER's attribute surface (py_envpool.h:303-332)
    for attr in ("_config_keys", "_default_config_values", "_state_keys", "_action_keys"):
        assert hasattr(ref_spec, attr)
    for attr in ("_send", "_recv", "_reset", "_render", "_xla", "_state_keys", "_action_keys"):
        assert hasattr(ref_pool, attr)
    nref = len(ref_spec._config_keys)
    # envpool_amd appends its extension keys (device, env_id_offset, recv_timeout_ms) AFTER the reference's
    assert list(our_spec._config_keys[:nref]) == list(ref_spec._config_keys)
    extra = list(our_spec._config_keys[nref:])  # (MuJoCo families: + precision)
    assert extra[-3:] == ["device", "env_id_offset", "recv_timeout_ms"]
    assert set(extra) <= {"precision", "device", "env_id_offset", "recv_timeout_ms"}
    assert _norm(our_spec._default_config_values[:nref]) == _norm(ref_spec._default_config_values)
    assert list(our_spec._state_keys) == list(ref_spec._state_keys)
    assert list(our_spec._action_keys) == list(ref_spec._action_keys)
    assert list(ref_pool._state_keys) == list(ref_spec._state_keys)
    # instantiated specs: dtype, shape, bounds, elementwise bounds, is_discrete per key
    conf = dict(zip(ref_spec._config_keys, ref_spec._default_config_values))
    conf.update(num_envs=8, seed=3)
    rs = ref_spec(tuple(conf[k] for k in ref_spec._config_keys))
    os_ = our_spec(tuple(conf[k] for k in ref_spec._config_keys)
                   + tuple(our_spec._default_config_values[nref:]))
    assert _norm(rs._config_values) == _norm(os_._config_values[:nref])
    assert _norm(rs._state_spec) == _norm(os_._state_spec)
    assert _norm(rs._action_spec) == _norm(os_._action_spec)
    # EnvSpec ctor check (env_spec.h:75-80) through pybind11: invalid_argument -> ValueError
    conf["batch_size"] = 9
    with pytest.raises(ValueError):
        ref_spec(tuple(conf[k] for k in ref_spec._config_keys))


def _actions(spec, rng, n):
    dtype, shape, bounds = spec._action_spec[-1][0], spec._action_spec[-1][1], spec._action_spec[-1][2]
    shape = [n] + [s for s in shape if s != -1]
    if np.issubdtype(dtype, np.integer):
        return rng.integers(bounds[0], bounds[1] + 1, size=shape).astype(dtype)
    return rng.uniform(bounds[0], bounds[1], size=shape).astype(dtype)


@pytest.mark.gpu
@pytest.mark.parametrize("stem,module,ours,task", FAMILIES, ids=[f[0] for f in FAMILIES])
def test_pybind_shim_is_bit_identical_to_the_ctypes_path(stem, module, ours, task):
    rb = _refbind()
    ref_spec = getattr(rb, f"_{stem}EnvSpec")
    ref_pool = getattr(rb, f"_{stem}EnvPool")
    our_spec, our_pool = _ours(module, ours)
    n = 96
    conf = dict(zip(ref_spec._config_keys, ref_spec._default_config_values))
    conf.update(num_envs=n, seed=11, max_episode_steps=40)
    if stem.startswith("Gym"):
        conf["post_constraint"] = False  # the v4 registration (mujoco/gym/registration.py:37-50)
    values = tuple(conf[k] for k in ref_spec._config_keys)
    rs = ref_spec(values)
    os_ = our_spec(values + tuple(our_spec._default_config_values[len(values):]))
    rp, op = ref_pool(rs), our_pool(os_)
    ids = np.arange(n, dtype=np.int32)
    rng = np.random.default_rng(0)

    def same(a, b, ctx):
        assert len(a) == len(b) == len(ref_spec._state_keys)
        for key, x, y in zip(ref_spec._state_keys, a, b):
            assert x.dtype == y.dtype and x.shape == y.shape, (ctx, key, x.dtype, y.dtype, x.shape, y.shape)
            assert np.array_equal(x, y, equal_nan=True), (ctx, key)

    rp._reset(ids)
    op._reset(ids)
    first = rp._recv()
    same(first, op._recv(), "reset")
    keep = [a.copy() for a in first]
    for t in range(60):
        act = _actions(rs, rng, n)
        for p in (rp, op):
            p._send([ids, ids, act])
        same(rp._recv(), op._recv(), t)
    # partial, permuted batch: rows in send order
    sub = rng.permutation(n)[:37].astype(np.int32)
    act = _actions(rs, rng, 37)
    for p in (rp, op):
        p._send([sub, sub, act])
    a, b = rp._recv(), op._recv()
    same(a, b, "partial")
    assert a[0].tolist() == sub.tolist()
    # arrays own their memory (capsule over the pinned block): never overwritten
    for x, y in zip(first, keep):
        assert np.array_equal(x, y, equal_nan=True)
    with pytest.raises(RuntimeError):
        rp._render(ids[:1], 64, 64, -1)
    with pytest.raises(ValueError):
        rp._reset(np.array([n + 3], dtype=np.int32))


def _same_tree(a, b, ctx):
    if isinstance(a, dict):
        assert set(a) == set(b), ctx
        for k in a:
            _same_tree(a[k], b[k], (ctx, k))
    else:
        assert np.array_equal(np.asarray(a), np.asarray(b)), ctx


@pytest.mark.gpu
def test_pybind_classes_slot_into_the_python_adaptors():
    """py_env(spec, pool) (envpool/python/api.py:22-41) over the pybind11 classes: the
    gymnasium adaptor steps HalfCheetah through the reference's PyEnvPool::PySend/PyRecv."""
    import envpool_amd as envpool
    from envpool_amd.python.api import py_env

    rb = _refbind()
    spec_cls, _dm, gym_cls = py_env(rb._GymHalfCheetahEnvSpec, rb._GymHalfCheetahEnvPool)
    n = 64
    conf = dict(zip(spec_cls._config_keys, spec_cls._default_config_values))
    conf.update(num_envs=n, seed=2, max_episode_steps=1000, post_constraint=False)
    env = gym_cls(spec_cls(tuple(conf[k] for k in spec_cls._config_keys)))
    ours = envpool.make_gym("HalfCheetah-v4", num_envs=n, seed=2)
    o1, i1 = env.reset()
    o2, i2 = ours.reset()
    assert np.array_equal(o1, o2) and set(i1) == set(i2)
    rng = np.random.default_rng(1)
    for _ in range(20):
        act = rng.uniform(-1, 1, (n, 6))
        r1, r2 = env.step(act), ours.step(act)
        for x, y in zip(r1[:4], r2[:4]):
            assert np.array_equal(x, y)
        _same_tree(r1[4], r2[4], "info")


@pytest.mark.gpu
def test_cc_driver_follows_the_reference_cc_test():
    exe = os.path.join(BUILD, "refbind_cc_test")
    if not os.path.exists(exe):
        pytest.skip("integration/_build/refbind_cc_test not built (needs /root/reference)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "refbind_cc_test: OK" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("stem,module,ours", [("GymHalfCheetah", "envpool_amd.mujoco.gym", "GymHalfCheetah"),
                                              ("GymHopper", "envpool_amd.mujoco.gym", "GymHopper")])
def test_pybind_shim_posts_its_block_at_send_for_big_batches(stem, module, ours):
    """`DeviceEnvPool::Send` of a whole sync pool names the batch's pinned block (epa_send_into) when the batch is big
    enough for it to matter; the step kernel writes the rows into it.  Same bytes as the ctypes path (which does the
    same from Python), arrays of earlier steps keep their values."""
    rb = _refbind()
    ref_spec, ref_pool = getattr(rb, f"_{stem}EnvSpec"), getattr(rb, f"_{stem}EnvPool")
    our_spec, our_pool = _ours(module, ours)
    n = 6000  # ~1 MB of results per batch: above the binding's posting threshold
    conf = dict(zip(ref_spec._config_keys, ref_spec._default_config_values))
    conf.update(num_envs=n, seed=5, max_episode_steps=7, post_constraint=False)
    values = tuple(conf[k] for k in ref_spec._config_keys)
    rs = ref_spec(values)
    rp, op = ref_pool(rs), our_pool(our_spec(values + tuple(our_spec._default_config_values[len(values):])))
    ids = np.arange(n, dtype=np.int32)
    rng = np.random.default_rng(2)
    rp._reset(ids), op._reset(ids)
    a, b = rp._recv(), op._recv()
    held = []
    for t in range(12):
        for x, y in zip(a, b):
            assert x.dtype == y.dtype and np.array_equal(x, y, equal_nan=True), t
        held.append(([x for x in a], [x.copy() for x in a]))
        act = _actions(rs, rng, n)
        rp._send([ids, ids, act]), op._send([ids, ids, act])
        a, b = rp._recv(), op._recv()
    for arrays, copies in held:
        for x, y in zip(arrays, copies):
            assert np.array_equal(x, y, equal_nan=True)
