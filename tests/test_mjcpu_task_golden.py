"""The plain-C restatement (oracle/mjcpu/tasks.c over engine.c, kind "port") replayed against
tests/golden/mujoco_task_<id>.npz -- rollouts of the reference's OWN gym-MuJoCo task wrappers
inside its own AsyncEnvPool (oracle/_ref/libref_mujoco.so; written by
tests/golden/make_mujoco_task_golden.py).  Same engine source under both, so BIT FOR BIT on every
state key; runs wherever the port oracle is built (no reference tree needed)."""
import glob
import os

import numpy as np
import pytest

from oracle import orc
from oracle.orc import Oracle

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "mujoco_task_*.npz")))

pytestmark = pytest.mark.skipif(not orc.have_port(), reason="oracle/_build/liboracle.so not built")


def test_fixture_set_is_complete():
    from tests.mj_util import GYM_VARIANTS
    names = {os.path.basename(p)[len("mujoco_task_"):-4] for p in GOLDEN}
    assert names == set(GYM_VARIANTS)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[12:-4] for p in GOLDEN])
def test_port_replays_the_reference_wrapper_rollout(path):
    g = np.load(path)
    acts = g["actions"]
    steps, n, _ = acts.shape
    port = Oracle(str(g["task"]), n, seed=int(g["seed"]), max_episode_steps=int(g["max_episode_steps"]),
                  extra=tuple(g["extra"]), kind="port")
    keys = [k[4:] for k in g.files if k.startswith("key:")]
    assert sorted(keys) == sorted(k for k, _, _ in port.keys)
    row = port.reset()
    for t in range(steps + 1):
        for k in keys:
            want = g["key:" + k][t]
            assert want.dtype == row[k].dtype and want.shape == row[k].shape, k
            assert np.array_equal(want.view(np.uint8), row[k].view(np.uint8)), (t, k)
        if t < steps:
            row = port.step(acts[t])
