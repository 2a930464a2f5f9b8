"""Pin of the MuJoCo arithmetic: oracle/mjcpu AND the HIP kernels against golden
vectors recorded from real MuJoCo 3.6.0 by tools/pin_with_mujoco.py
(tests/golden/mujoco_*.npz).  The reference pins the same boundary with
envpool/mujoco/gym/mujoco_gym_align_test.py:120-171 (obs atol 1e-6 / rtol 1e-7
against mujoco 3.6.0); those tolerances are used here.

MuJoCo 3.6.0 is reachable from neither the build container nor the GPU boxes
(profiles/archive/r2_probe_*.log, profiles/r4_probe_container.log, profiles/r4z_probe_gpu_box.log), so until somebody runs the pin script on a machine
that has the wheel the golden files do not exist, the `*_real_mujoco` tests
SKIP, and MuJoCo parity stays UNPINNED.  Dropping the .npz files into
tests/golden/ flips every test below without a code change; the `*_harness_*`
tests run the very same checkers on stand-in records produced by oracle/mjcpu
itself so that the plumbing (state injection, field order, tolerances) is known
to work -- they pin nothing.
"""
import ctypes
import os

import numpy as np
import pytest

from mj_util import RawMj

GOLD = os.path.join(os.path.dirname(__file__), "golden")

# golden file stem, oracle task, HIP family, HIP params
MODELS = [
    ("half_cheetah", "HalfCheetah", "HalfCheetah", {}),
    ("ant", "Ant", "Ant", {"post_constraint": 1, "use_contact_force": 1}),
    ("walker2d", "Walker2d", "Walker2d", {}),
    ("walker2d_v5", "Walker2dV5", "Walker2d", {"xml_v5": 1, "legacy_healthy_reward": 0}),
    ("inverted_pendulum", "InvertedPendulum", "InvertedPendulum", {}),
    ("inverted_double_pendulum", "InvertedDoublePendulum", "InvertedDoublePendulum", {}),
    ("reacher", "Reacher", "Reacher", {}),
    ("swimmer", "Swimmer", "Swimmer", {}),
    ("hopper", "Hopper", "Hopper", {}),
    ("humanoid", "Humanoid", "Humanoid", {"post_constraint": 1}),
    ("humanoidstandup", "HumanoidStandup", "HumanoidStandup", {"post_constraint": 1}),
]
IDS = [m[0] for m in MODELS]
# the reference's own alignment tolerance (mujoco_gym_align_test.py:38-80)
ATOL, RTOL = 1e-6, 1e-7


def _load(name):
    path = os.path.join(GOLD, f"mujoco_{name}.npz")
    if not os.path.exists(path):
        pytest.skip("no golden vectors from real MuJoCo (run tools/pin_with_mujoco.py)")
    return np.load(path)


# ---------------------------------------------------------------------------
# checkers (shared by the real-MuJoCo tests and the harness self-checks)
# ---------------------------------------------------------------------------
def check_oracle_steps(g, task, stride=7):
    o = RawMj(task)
    if "body_mass" in g:
        np.testing.assert_allclose(o.body_mass, g["body_mass"], rtol=1e-9)
        np.testing.assert_allclose(o.dof_invweight0, g["dof_invweight0"], rtol=1e-7)
    fs = int(g["frame_skip"]) if "frame_skip" in g else 5
    for i in range(0, len(g["qpos0"]), stride):
        # the recorded state includes qacc_warmstart (the unconverged PGS of the humanoids
        # depends on it); no forward pass in between, like the recording
        o.set_warm(g["qpos0"][i], g["qvel0"][i], g["ctrl"][i], g["warm0"][i])
        o.step(fs)
        q, v, _ = o.get()
        np.testing.assert_allclose(q, g["qpos1"][i], atol=ATOL, rtol=RTOL, err_msg=f"qpos@{i}")
        np.testing.assert_allclose(v, g["qvel1"][i], atol=ATOL, rtol=RTOL, err_msg=f"qvel@{i}")
        if "cinert1" in g:  # fields of the last forward evaluation that Humanoid observes
            cinert, cvel, qfrc_act, cfrc = o.observed()
            np.testing.assert_allclose(cinert, g["cinert1"][i], atol=ATOL, rtol=RTOL)
            np.testing.assert_allclose(cvel, g["cvel1"][i], atol=ATOL, rtol=RTOL)
            np.testing.assert_allclose(qfrc_act, g["qfrc_actuator1"][i], atol=ATOL, rtol=RTOL)
            np.testing.assert_allclose(cfrc, g["cfrc_ext1"][i], atol=1e-5, rtol=1e-6)


def oracle_stage(o, name, cap=1 << 16):
    out = np.zeros(cap)
    fn = o.L.mjcpu_raw_stage
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int]
    fn.restype = ctypes.c_int
    k = fn(o.inner, 0, name.encode(), out.ctypes.data, cap)
    assert 0 <= k <= cap, (name, k)
    return out[:k].copy()


# pipeline order: the first failing entry names the stage
STAGE_ORDER = [
    ("M1 kinematics", ["xpos", "xquat", "xipos", "subtree_com", "cinert", "cdof"]),
    ("M2 crb", ["qM"]),
    ("M5 fwdVelocity", ["cvel", "qfrc_passive", "qfrc_bias"]),
    ("M6 fwdActuation", ["qfrc_actuator"]),
    ("M7 fwdAcceleration", ["qacc_smooth"]),
]
EFC_FIELDS = ["efc_pos", "efc_margin", "efc_vel", "efc_diagApprox", "efc_R", "efc_D", "efc_aref"]


def check_oracle_stages(g, task):
    """One mj_forward at each staged sample; fields compared in pipeline order.  Constraint
    rows are matched by their Jacobian row (the order of contacts inside one geom pair may
    legitimately differ), so a mismatch reads "M4 makeConstraint: efc_aref of row ..."."""
    o = RawMj(task)
    samples = sorted({int(k.split("/")[1]) for k in g.files if k.startswith("stage/")})
    assert samples, "golden file carries no staged fields (old pin script?)"
    o.L.mjcpu_raw_forward.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for i in samples:
        f = lambda name: g[f"stage/{i}/{name}"]  # noqa: E731
        o.set_warm(g["qpos0"][i], g["qvel0"][i], np.zeros(o.nu), g["warm0"][i])
        o.L.mjcpu_raw_forward(o.inner, 0)
        for stage, fields in STAGE_ORDER:
            for name in fields:
                np.testing.assert_allclose(oracle_stage(o, name), f(name), atol=ATOL, rtol=RTOL,
                                           err_msg=f"{stage}: {name} @ sample {i}")
        ncon, nefc, _ = (int(x) for x in f("counts"))
        mine = oracle_stage(o, "counts")
        assert (int(mine[0]), int(mine[1])) == (ncon, nefc), f"M3 collision: ncon/nefc @ sample {i}"
        # M3: contacts as a set keyed by (geom1, geom2, pos)
        gc = f("contact").reshape(ncon, 18)
        mc = oracle_stage(o, "contact").reshape(ncon, 18)
        for row in gc:
            d = np.abs(mc[:, :2] - row[:2]).sum(axis=1) + np.abs(mc[:, 4:7] - row[4:7]).sum(axis=1)
            j = int(np.argmin(d))
            np.testing.assert_allclose(mc[j, :16], row[:16], atol=ATOL, rtol=RTOL,
                                       err_msg=f"M3 collision: contact {row[:2]} @ sample {i}")
        # M4: rows matched by Jacobian
        nv = o.nv
        gj, mj_ = f("efc_J").reshape(nefc, nv), oracle_stage(o, "efc_J").reshape(nefc, nv)
        match = [int(np.argmin(np.abs(mj_ - r).sum(axis=1))) for r in gj]
        assert sorted(match) == list(range(nefc)), f"M4 makeConstraint: efc_J rows @ sample {i}"
        np.testing.assert_allclose(mj_[match], gj, atol=ATOL, rtol=RTOL,
                                   err_msg=f"M4 makeConstraint: efc_J @ sample {i}")
        for name in EFC_FIELDS:
            np.testing.assert_allclose(oracle_stage(o, name)[match], f(name), atol=ATOL, rtol=RTOL,
                                       err_msg=f"M4 makeConstraint: {name} @ sample {i}")
        kbi = oracle_stage(o, "efc_KBIP").reshape(nefc, 3)[match]
        np.testing.assert_allclose(kbi, f("efc_KBIP").reshape(nefc, 4)[:, :3], atol=ATOL, rtol=RTOL,
                                   err_msg=f"M4 makeConstraint: efc_KBIP @ sample {i}")
        # M8: solver result (Newton: unique minimiser; PGS: warm start + sweep order matter)
        np.testing.assert_allclose(oracle_stage(o, "qacc"), f("qacc"), atol=1e-5, rtol=1e-6,
                                   err_msg=f"M8 solver: qacc @ sample {i}")
        np.testing.assert_allclose(oracle_stage(o, "efc_force")[match], f("efc_force"),
                                   atol=1e-5, rtol=1e-6, err_msg=f"M8 solver: efc_force @ sample {i}")


def check_hip_steps(g, family, params, stride=3):
    """The GPU leg: inject (qpos0, qvel0, qacc_warmstart) into the HIP pool through the C
    ABI's state hook, step once with the recorded ctrl, read the state back and compare with
    MuJoCo's (qpos1, qvel1); Humanoid: also the observed cinert / cvel / qfrc_actuator /
    cfrc_ext.  One env per golden sample, one launch for the whole file."""
    from envpool_amd.core.device_pool import DevicePool

    sel = np.arange(0, len(g["qpos0"]), stride)
    n = len(sel)
    nq, nv = g["qpos0"].shape[1], g["qvel0"].shape[1]
    pool = DevicePool(family, n, seed=0, max_episode_steps=100000,
                      params={"precision": 1, **params} if family in
                      ("HalfCheetah", "Ant", "Walker2d", "Hopper") else params)
    ids = np.arange(n, dtype=np.int32)
    pool.reset(ids)
    pool.recv()
    st = pool.get_state()
    st[:, :nq] = g["qpos0"][sel]
    st[:, nq:nq + nv] = g["qvel0"][sel]
    st[:, nq + nv:nq + 2 * nv] = g["warm0"][sel]
    pool.set_state(st)
    pool.send(ids, np.ascontiguousarray(g["ctrl"][sel]))
    out = pool.recv_dict()
    st1 = pool.get_state()
    # terminated envs still hold the stepped state (the reset happens on the NEXT step)
    np.testing.assert_allclose(st1[:, :nq], g["qpos1"][sel], atol=ATOL, rtol=RTOL, err_msg="qpos1")
    np.testing.assert_allclose(st1[:, nq:nq + nv], g["qvel1"][sel], atol=ATOL, rtol=RTOL,
                               err_msg="qvel1")
    if family.startswith("Humanoid"):
        obs = out["obs"]
        nb = g["cinert1"].shape[1]
        o = (nq - 2) + nv
        for name, width, atol, rtol in (("cinert1", nb * 10, ATOL, RTOL), ("cvel1", nb * 6, ATOL, RTOL),
                                        ("qfrc_actuator1", nv, ATOL, RTOL),
                                        ("cfrc_ext1", nb * 6, 1e-5, 1e-6)):
            want = g[name][sel].reshape(n, -1)
            np.testing.assert_allclose(obs[:, o:o + width], want, atol=atol, rtol=rtol, err_msg=name)
            o += width
        assert o == obs.shape[1]
    return n


def check_reset_frame(g, task):
    """What mj_forward leaves in qpos at a reset (un-normalised free-joint quaternion)."""
    o = RawMj(task)
    for qin, qout in zip(g["reset_qpos_in"][:8], g["reset_qpos_out"][:8]):
        o.set(qin, np.zeros(o.nv))  # reset + mj_forward
        q, _, _ = o.get()
        np.testing.assert_allclose(q, qout, atol=1e-15, rtol=0)


# ---------------------------------------------------------------------------
# real MuJoCo (skip until tests/golden/mujoco_*.npz exist)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name,task,family,params", MODELS, ids=IDS)
def test_oracle_matches_real_mujoco(name, task, family, params):
    check_oracle_steps(_load(name), task)


@pytest.mark.parametrize("name,task,family,params", MODELS, ids=IDS)
def test_oracle_stages_match_real_mujoco(name, task, family, params):
    g = _load(name)
    check_oracle_stages(g, task)
    check_reset_frame(g, task)


@pytest.mark.gpu
@pytest.mark.parametrize("name,task,family,params", MODELS, ids=IDS)
def test_hip_kernels_match_real_mujoco(name, task, family, params):
    check_hip_steps(_load(name), family, params)


# ---------------------------------------------------------------------------
# harness self-checks on stand-in records made by the oracle (pin NOTHING)
# ---------------------------------------------------------------------------
class _Rec(dict):
    @property
    def files(self):
        return list(self.keys())


def standin_record(task, amax, frame_skip, samples=24, stage_samples=(0, 1, 2), seed=0):
    """Same layout as tools/pin_with_mujoco.py writes, produced by oracle/mjcpu."""
    from oracle.orc import Oracle

    o = RawMj(task)
    o.L.mjcpu_raw_forward.argtypes = [ctypes.c_void_p, ctypes.c_int]
    rng = np.random.default_rng(seed)
    rec = {k: [] for k in ("qpos0", "qvel0", "warm0", "ctrl", "qpos1", "qvel1", "cinert1", "cvel1",
                           "qfrc_actuator1", "cfrc_ext1")}
    stages = {}
    # a valid start state: the task's own reset (init_qpos + noise)
    t0 = Oracle(task, 1, seed=seed, max_episode_steps=1000)
    t0.reset()
    st = t0.get_state()[0]
    qpos, qvel = st[:o.nq].copy(), st[o.nq:o.nq + o.nv].copy()
    warm = np.zeros(o.nv)
    for i in range(samples):
        ctrl = rng.uniform(-amax, amax, o.nu)
        if i in stage_samples:
            o.set_warm(qpos, qvel, np.zeros(o.nu), warm)
            o.L.mjcpu_raw_forward(o.inner, 0)
            for _, fields in STAGE_ORDER:
                for name in fields:
                    stages[f"stage/{i}/{name}"] = oracle_stage(o, name)
            for name in EFC_FIELDS + ["efc_J", "efc_force", "qacc", "contact", "counts"]:
                stages[f"stage/{i}/{name}"] = oracle_stage(o, name)
            kbi = oracle_stage(o, "efc_KBIP").reshape(-1, 3)
            stages[f"stage/{i}/efc_KBIP"] = np.concatenate([kbi, np.zeros((len(kbi), 1))], 1).ravel()
        o.set_warm(qpos, qvel, ctrl, warm)
        rec["qpos0"].append(qpos.copy()), rec["qvel0"].append(qvel.copy())
        rec["warm0"].append(warm.copy()), rec["ctrl"].append(ctrl)
        o.step(frame_skip)
        warm = oracle_warm(o)
        qpos, qvel, _ = o.get()
        cinert, cvel, qact, cfrc = o.observed()
        rec["qpos1"].append(qpos.copy()), rec["qvel1"].append(qvel.copy())
        rec["cinert1"].append(cinert), rec["cvel1"].append(cvel)
        rec["qfrc_actuator1"].append(qact), rec["cfrc_ext1"].append(cfrc)
    out = _Rec({k: np.array(v) for k, v in rec.items()})
    out.update(stages)
    out["frame_skip"] = np.array(frame_skip)
    return out


def oracle_warm(o):
    """qacc_warmstart of the oracle's env 0 (through the flat task state)."""
    st = np.zeros(4096)
    fn = o.L.mjcpu_get_state
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    ids = np.zeros(1, dtype=np.int32)
    fn(o.inner, ids.ctypes.data, 1, st.ctypes.data)
    return st[o.nq + o.nv:o.nq + 2 * o.nv].copy()


_HARNESS = [("half_cheetah", 1.0, 5), ("ant", 1.0, 5), ("hopper", 1.0, 4), ("humanoid", 0.4, 5)]


@pytest.mark.parametrize("name,amax,fs", _HARNESS, ids=[h[0] for h in _HARNESS])
def test_harness_oracle_checkers_run(name, amax, fs):
    _, task, _, _ = MODELS[IDS.index(name)]
    g = standin_record(task, amax, fs, samples=8)
    check_oracle_steps(g, task, stride=1)
    check_oracle_stages(g, task)


@pytest.mark.gpu
@pytest.mark.parametrize("name,amax,fs", _HARNESS + [("reacher", 1.0, 2), ("swimmer", 1.0, 4),
                                                     ("inverted_double_pendulum", 1.0, 5),
                                                     ("walker2d_v5", 1.0, 4), ("humanoidstandup", 0.4, 5)],
                         ids=lambda v: v if isinstance(v, str) else None)
def test_harness_hip_checker_runs(name, amax, fs):
    """The GPU leg on stand-in records: state injection by (qpos, qvel, warm start), one
    env-step, state read-back -- the exact code path the real golden files will take."""
    _, task, family, params = MODELS[IDS.index(name)]
    g = standin_record(task, amax, fs, samples=24, stage_samples=())
    assert check_hip_steps(g, family, params, stride=1) == 24
