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
with a GPU, the HIP kernels: test_hip_kernels_match_real_mujoco) against them
with the reference's own tolerance (obs atol 1e-6, rtol 1e-7:
envpool/mujoco/gym/mujoco_gym_align_test.py:38-80).

Besides the (state, ctrl) -> next-state vectors it records, for the first
samples of every episode, MuJoCo's per-stage fields of ONE mj_forward at the
recorded state (qM, cinert, cdof, qfrc_bias, qfrc_passive, contacts, efc_J,
efc_aref, efc_R, efc_D, efc_KBIP, qacc_smooth, qacc, efc_force ...): when a
golden step does not match, test_oracle_stages_match_real_mujoco says which
pipeline stage is off.  It also records what mj_forward
leaves in qpos for an un-normalised free-joint quaternion (the reset frame of
Ant / Humanoid: MuJoCo >= 3.1.4 does not normalise qpos in place).
"""
import copy
import os
import sys

import numpy as np


STAGE_FIELDS = ("xpos", "xquat", "xipos", "subtree_com", "cinert", "cdof", "cvel", "qfrc_passive",
                "qfrc_bias", "qfrc_actuator", "qacc_smooth", "qacc", "qfrc_constraint", "efc_pos",
                "efc_margin", "efc_vel", "efc_aref", "efc_R", "efc_D", "efc_diagApprox", "efc_KBIP",
                "efc_force")


def dump_stages(mujoco, m, d) -> dict:
    """One mj_forward on a COPY of `d` (the trajectory is not disturbed); every field as a
    flat float64 array under MuJoCo's own member name."""
    d2 = copy.copy(d)
    mujoco.mj_forward(m, d2)
    out = {}
    full = np.zeros((m.nv, m.nv))
    mujoco.mj_fullM(m, full, d2.qM)
    out["qM"] = full.ravel()
    for f in STAGE_FIELDS:
        out[f] = np.asarray(getattr(d2, f), dtype=np.float64).ravel().copy()
    assert not mujoco.mj_isSparse(m), "dense Jacobian expected (nv < 60)"
    out["efc_J"] = np.asarray(d2.efc_J, dtype=np.float64).ravel()[: d2.nefc * m.nv].copy()
    con = []
    for c in d2.contact:
        con += [c.geom[0], c.geom[1], c.dist, c.includemargin, *c.pos, *c.frame, c.dim, c.efc_address]
    out["contact"] = np.asarray(con, dtype=np.float64)
    out["counts"] = np.asarray([d2.ncon, d2.nefc, int(d2.solver_niter[0])], dtype=np.float64)
    return out


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
        stages = {}
        reset_in, reset_out = [], []
        for ep in range(8 * 200 // horizon):
            mujoco.mj_resetData(m, d)
            d.qpos[:] = m.qpos0 + rng.uniform(-0.1, 0.1, m.nq)
            d.qvel[:] = rng.normal(0, 0.1, m.nv)
            if name == "reacher":
                d.qvel[2:] = 0.0  # the target never moves (reacher.h:116-131)
            reset_in.append(d.qpos.copy())
            mujoco.mj_forward(m, d)
            reset_out.append(d.qpos.copy())  # == reset_in unless mj_forward normalises in place
            for t in range(horizon):
                if t < 3 or t == horizon - 1:  # per-stage fields at this (state, warm start)
                    d.ctrl[:] = 0.0
                    for f, v in dump_stages(mujoco, m, d).items():
                        stages[f"stage/{len(rec['qpos0'])}/{f}"] = v
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
        extra = dict(frame_skip=frame_skip, reset_qpos_in=np.array(reset_in),
                     reset_qpos_out=np.array(reset_out), mujoco_version=np.array(mujoco.__version__),
                     **stages, body_mass=m.body_mass.copy(), dof_invweight0=m.dof_invweight0.copy(),
                     body_invweight0=m.body_invweight0.copy())
        np.savez_compressed(os.path.join(out_dir, f"mujoco_{name}.npz"),
                            **{k: np.array(v) for k, v in rec.items()}, **extra)
        print("wrote", name)


if __name__ == "__main__":
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    main(sys.argv[1])
