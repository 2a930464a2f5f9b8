"""oracle/mjcpu/tasks.c (the plain-C restatement of the gym task wrappers) against the
reference's OWN wrappers: envpool/mujoco/gym/*.h compiled in place inside the reference's own
AsyncEnvPool (oracle/_ref/libref_mujoco.so, oracle/ref_mujoco_driver.cc) over a mujoco.h shim
whose engine calls forward to oracle/mjcpu.  Same engine object code underneath both, so every
state key must agree BIT FOR BIT: this pins the reset draw order (libstdc++'s real
uniform_real_distribution / normal_distribution), rewards, healthy / termination rules,
observation and info assembly, post_constraint and the done / trunc / elapsed_step bookkeeping
of the restatement to reference code.  It does not pin the engine arithmetic.

Calling sequence as in the reference's own C++ test
(envpool/mujoco/gym/mujoco_gym_envpool_test.cc:27-56)."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import orc
from oracle.orc import Oracle
from tests.mj_util import GYM_VARIANTS, mj_extra

pytestmark = pytest.mark.skipif(not (orc.have_ref_mujoco() and orc.have_port()),
                                reason="oracle/_ref/libref_mujoco.so not built (no /root/reference)")


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


def _run_pair(task, max_steps, extra, n, steps, seed, threads=3):
    port = Oracle(task, n, seed=seed, max_episode_steps=max_steps, extra=extra, kind="port")
    ref = Oracle(task, n, seed=seed, max_episode_steps=max_steps, extra=extra,
                 kind="reference_mujoco", num_threads=threads)
    assert ref.kind == "reference_mujoco"
    assert port.keys == ref.keys
    assert (port.action_dtype, port.action_elems) == (ref.action_dtype, ref.action_elems)
    rng = np.random.default_rng(seed)
    a, b = port.reset(), ref.reset()
    dones = resets = 0
    for t in range(steps + 1):
        for k in a:
            assert np.array_equal(_bits(a[k]), _bits(b[k])), (task, t, k, a[k].ravel()[:8], b[k].ravel()[:8])
        dones += int(a["done"].sum())
        resets += int((a["elapsed_step"] == 0).sum())
        act = rng.uniform(-1.0, 1.0, size=(n, port.action_elems))
        a, b = port.step(act), ref.step(act)
    return dones, resets


@pytest.mark.parametrize("name", sorted(GYM_VARIANTS))
def test_task_restatement_is_the_reference_wrapper_bit_for_bit(name):
    task, max_steps, over = GYM_VARIANTS[name]
    dones, resets = _run_pair(task, max_steps, mj_extra(task, **over), n=6, steps=320, seed=11)
    # every task either terminates or truncates inside 320 steps except the 1000-step ones that
    # cannot fall; the auto-reset rows (elapsed_step == 0 after the initial reset) are compared too
    if max_steps <= 100 or task in ("Walker2d", "Walker2dV5", "Hopper", "InvertedPendulum",
                                    "InvertedDoublePendulum", "Humanoid"):
        assert dones > 0 and resets > 6


@pytest.mark.parametrize("name", ["HalfCheetah-v4", "Ant-v4", "Humanoid-v4"])
def test_short_episodes_truncate_like_the_reference(name):
    task, _, over = GYM_VARIANTS[name]
    dones, resets = _run_pair(task, 7, mj_extra(task, **over), n=5, steps=40, seed=2)
    assert dones >= 5 * 5


@pytest.mark.parametrize("name,over", [
    ("HalfCheetah-v4", dict(ctrl_cost_weight=0.3, forward_reward_weight=2.0, reset_noise_scale=0.03, frame_skip=3)),
    ("Ant-v4", dict(ctrl_cost_weight=0.1, reset_noise_scale=0.2, use_contact_force=1, post_constraint=1)),
    ("Pusher-v4", dict(dist_cost_weight=2.0, near_cost_weight=0.25, weighted_reward_info=1)),
    ("Hopper-v4", dict(legacy_healthy_reward=0, frame_skip=2)),
])
def test_non_default_options(name, over):
    task, max_steps, base = GYM_VARIANTS[name]
    _run_pair(task, max_steps, mj_extra(task, **{**base, **over}), n=4, steps=150, seed=5)


def test_partial_id_sends_and_reset_subsets():
    """Send() for a subset of ids and Reset(ids) mid-episode (mujoco_gym_envpool_test.cc shape)."""
    task, ms, over = GYM_VARIANTS["Walker2d-v4"]
    ex = mj_extra(task, **over)
    n = 8
    port = Oracle(task, n, seed=9, max_episode_steps=ms, extra=ex, kind="port")
    ref = Oracle(task, n, seed=9, max_episode_steps=ms, extra=ex, kind="reference_mujoco", num_threads=2)
    rng = np.random.default_rng(1)
    a, b = port.reset(), ref.reset()
    for t in range(60):
        ids = np.sort(rng.choice(n, size=int(rng.integers(1, n + 1)), replace=False)).astype(np.int32)
        act = rng.uniform(-1, 1, size=(len(ids), port.action_elems))
        if t % 17 == 5:
            a, b = port.reset(ids), ref.reset(ids)
        else:
            a, b = port.step(act, ids), ref.step(act, ids)
        for k in a:
            assert np.array_equal(_bits(a[k]), _bits(b[k])), (t, k)


def test_frame_stack_is_the_reference_ring():
    """frame_stack=3 through the reference's own FrameStackBuffer (envpool/mujoco/frame_stack.h)
    equals stacking the restatement's single frames: reset fills every slot, a step shifts."""
    task, ms, over = GYM_VARIANTS["HalfCheetah-v4"]
    n, fs = 3, 3
    port = Oracle(task, n, seed=4, max_episode_steps=9, extra=mj_extra(task, **over), kind="port")
    ref = Oracle(task, n, seed=4, max_episode_steps=9, extra=mj_extra(task, frame_stack=fs, **over),
                 kind="reference_mujoco")
    assert dict((k, e) for k, _, e in ref.keys)["obs"] == fs * 17
    rng = np.random.default_rng(0)
    a, b = port.reset(), ref.reset()
    ring = np.repeat(a["obs"][:, None, :], fs, axis=1)
    for t in range(30):
        assert np.array_equal(_bits(ring.reshape(n, -1)), _bits(b["obs"])), t
        act = rng.uniform(-1, 1, size=(n, 6))
        a, b = port.step(act), ref.step(act)
        for e in range(n):
            if a["elapsed_step"][e, 0] == 0:
                ring[e] = a["obs"][e]
            else:
                ring[e] = np.concatenate([ring[e, 1:], a["obs"][e][None]], axis=0)


REF_XML = "/root/reference/third_party/mujoco_gym_xml_patches"


@pytest.mark.skipif(not os.path.isdir(REF_XML), reason="no reference tree")
@pytest.mark.parametrize("xml,bodies", [
    ("ant", ["torso"]),
    ("reacher", ["fingertip", "target"]),
    ("pusher", ["tips_arm", "object", "goal"]),
    ("pusher_v5", ["tips_arm", "object", "goal"]),
])
def test_body_ids_are_the_xml_document_order(xml, bodies):
    """mj_name2id of the shim == MuJoCo's rule (body ids in document order, world = 0),
    re-derived from the reference's XML."""
    text = open(os.path.join(REF_XML, xml + "_envpool.xml")).read()
    text = re.sub(r"<!--.*?-->", "", text, flags=re.S)
    order = [m.group(1) or "" for m in re.finditer(r"<body\b(?:[^>]*?\bname=\"([^\"]*)\")?[^>]*>", text)]
    lib = ctypes.CDLL(orc.REF_MUJOCO_LIB)
    lib.ref_mujoco_body_id.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    assert lib.ref_mujoco_nbody((xml + ".xml").encode()) == len(order) + 1
    for b in bodies:
        assert lib.ref_mujoco_body_id((xml + ".xml").encode(), b.encode()) == order.index(b) + 1
