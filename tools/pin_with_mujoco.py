"""Dump golden (state, action) -> next-state vectors from REAL MuJoCo.

oracle/mjcpu is a restatement of MuJoCo 3.6.0's pipeline; neither MuJoCo nor
the reference's XML assets can be used in the build container (no `mujoco`
wheel, no network), so its parity is unpinned.  Run this script on ANY machine
that has `mujoco==3.6.0` and a checkout of the reference:

    python tools/pin_with_mujoco.py /path/to/envpool/third_party/mujoco_gym_xml_patches

It writes tests/golden/mujoco_{half_cheetah,ant,walker2d,walker2d_v5,
inverted_pendulum,inverted_double_pendulum,reacher,swimmer,hopper,humanoid,
humanoidstandup}.npz; tests/test_mjcpu_golden.py
activates automatically when those files exist and checks oracle/mjcpu (and,
with a GPU, the HIP kernels) against them with the reference's own tolerance
(obs atol 1e-6, rtol 1e-7: envpool/mujoco/gym/mujoco_gym_align_test.py:38-80).
"""
import os
import sys

import numpy as np


def main(xml_dir: str) -> None:
    import mujoco  # noqa: PLC0415

    assert mujoco.__version__.startswith("3.6"), mujoco.__version__
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                           "tests", "golden")
    rng = np.random.default_rng(2024)
    # name, xml, frame_skip, action range, episode length before a re-randomised reset
    for name, xml, frame_skip, amax, horizon in (
            ("half_cheetah", "half_cheetah_envpool.xml", 5, 1.0, 200),
            ("ant", "ant_envpool.xml", 5, 1.0, 200),
            ("walker2d", "walker2d_envpool.xml", 4, 1.0, 40),
            ("walker2d_v5", "walker2d_v5_envpool.xml", 4, 1.0, 40),
            ("inverted_pendulum", "inverted_pendulum_envpool.xml", 2, 3.0, 25),
            ("inverted_double_pendulum", "inverted_double_pendulum_envpool.xml", 5, 1.0, 25),
            ("reacher", "reacher_envpool.xml", 2, 1.0, 50),
            ("swimmer", "swimmer_envpool.xml", 4, 1.0, 200),
            ("hopper", "hopper_envpool.xml", 4, 1.0, 25),
            ("humanoid", "humanoid_envpool.xml", 5, 0.4, 25),
            ("humanoidstandup", "humanoidstandup_envpool.xml", 5, 0.4, 100)):
        m = mujoco.MjModel.from_xml_path(os.path.join(xml_dir, xml))
        d = mujoco.MjData(m)
        rec = {k: [] for k in ("qpos0", "qvel0", "warm0", "ctrl", "qpos1", "qvel1", "xpos1",
                               "qfrc_constraint1", "cfrc_ext1", "cinert1", "cvel1",
                               "qfrc_actuator1", "xipos1", "solver_niter1", "nefc1")}
        for ep in range(8 * 200 // horizon):
            mujoco.mj_resetData(m, d)
            d.qpos[:] = m.qpos0 + rng.uniform(-0.1, 0.1, m.nq)
            d.qvel[:] = rng.normal(0, 0.1, m.nv)
            mujoco.mj_forward(m, d)
            for t in range(horizon):
                rec["qpos0"].append(d.qpos.copy())
                rec["qvel0"].append(d.qvel.copy())
                rec["warm0"].append(d.qacc_warmstart.copy())
                ctrl = rng.uniform(-amax, amax, m.nu)
                rec["ctrl"].append(ctrl)
                d.ctrl[:] = ctrl
                for _ in range(frame_skip):
                    mujoco.mj_step(m, d)
                rec["qpos1"].append(d.qpos.copy())
                rec["qvel1"].append(d.qvel.copy())
                rec["xpos1"].append(d.xpos[1].copy())
                # lagged mjData fields the tasks observe (last RK4 stage) and cfrc_ext
                rec["qfrc_constraint1"].append(d.qfrc_constraint.copy())
                # what the Humanoid tasks observe (humanoid.h:229-257) and the solver's trace
                rec["cinert1"].append(d.cinert.copy())
                rec["cvel1"].append(d.cvel.copy())
                rec["qfrc_actuator1"].append(d.qfrc_actuator.copy())
                rec["xipos1"].append(d.xipos.copy())
                rec["solver_niter1"].append(int(d.solver_niter[0]))
                rec["nefc1"].append(int(d.nefc))
                mujoco.mj_rnePostConstraint(m, d)
                rec["cfrc_ext1"].append(d.cfrc_ext.copy())
        extra = dict(frame_skip=frame_skip, body_mass=m.body_mass.copy(), dof_invweight0=m.dof_invweight0.copy(),
                     body_invweight0=m.body_invweight0.copy())
        np.savez_compressed(os.path.join(out_dir, f"mujoco_{name}.npz"),
                            **{k: np.array(v) for k, v in rec.items()}, **extra)
        print("wrote", name)


if __name__ == "__main__":
    main(sys.argv[1])
