"""Activates when tools/pin_with_mujoco.py has produced golden vectors from
real MuJoCo 3.6.0 (tests/golden/mujoco_*.npz); until then MuJoCo parity of
oracle/mjcpu is UNPINNED and this file skips."""
import os

import numpy as np
import pytest

from mj_util import RawMj

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name,task", [
    ("half_cheetah", "HalfCheetah"), ("ant", "Ant"), ("walker2d", "Walker2d"),
    ("walker2d_v5", "Walker2dV5"), ("inverted_pendulum", "InvertedPendulum"),
    ("inverted_double_pendulum", "InvertedDoublePendulum"), ("reacher", "Reacher"), ("swimmer", "Swimmer"), ("hopper", "Hopper"),
    ("humanoid", "Humanoid"), ("humanoidstandup", "HumanoidStandup")])
def test_oracle_matches_real_mujoco(name, task):
    path = os.path.join(GOLD, f"mujoco_{name}.npz")
    if not os.path.exists(path):
        pytest.skip("no golden vectors from real MuJoCo (run tools/pin_with_mujoco.py)")
    g = np.load(path)
    o = RawMj(task)
    np.testing.assert_allclose(o.body_mass, g["body_mass"], rtol=1e-9)
    np.testing.assert_allclose(o.dof_invweight0, g["dof_invweight0"], rtol=1e-7)
    for i in range(0, len(g["qpos0"]), 7):
        # the recorded state includes qacc_warmstart (the unconverged PGS of the humanoids
        # depends on it); no forward pass in between, like the recording
        o.set_warm(g["qpos0"][i], g["qvel0"][i], g["ctrl"][i], g["warm0"][i])
        o.step(int(g["frame_skip"]) if "frame_skip" in g else 5)
        q, v, _ = o.get()
        # the reference's own alignment tolerance (mujoco_gym_align_test.py:38-80)
        np.testing.assert_allclose(q, g["qpos1"][i], atol=1e-6, rtol=1e-7)
        np.testing.assert_allclose(v, g["qvel1"][i], atol=1e-6, rtol=1e-7)
        if "cinert1" in g:  # fields of the last forward evaluation that Humanoid observes
            cinert, cvel, qfrc_act, cfrc = o.observed()
            np.testing.assert_allclose(cinert, g["cinert1"][i], atol=1e-6, rtol=1e-7)
            np.testing.assert_allclose(cvel, g["cvel1"][i], atol=1e-6, rtol=1e-7)
            np.testing.assert_allclose(qfrc_act, g["qfrc_actuator1"][i], atol=1e-6, rtol=1e-7)
            np.testing.assert_allclose(cfrc, g["cfrc_ext1"][i], atol=1e-5, rtol=1e-6)
