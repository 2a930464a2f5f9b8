"""Physics invariants of oracle/mjcpu that need no external oracle
(SURVEY Appendix A.10).  Real-MuJoCo parity of this restatement is UNPINNED;
these tests pin its internal consistency, and tests/cpu_harness cross-checks it
against the independently formulated product code."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from mj_util import _H, RawMj

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ANT_Q0 = np.array([0, 0, 0.75, 1, 0, 0, 0, 0, 0.9, 0, -0.9, 0, -0.9, 0, 0.9], float)
# extra = [frame_skip, ctrl_w, fwd_w, noise, no_contact, no_limit, no_act, no_passive, integrator, h]
FREE = (5, 0.5, 1, 0.1, 1, 1, 1, 1)


def test_model_compile_constants():
    c, a = RawMj("HalfCheetah"), RawMj("Ant")
    assert (c.nq, c.nv, c.nu, c.nbody, c.ngeom) == (9, 9, 6, 8, 9)
    assert (a.nq, a.nv, a.nu, a.nbody, a.ngeom) == (15, 14, 8, 14, 14)
    assert abs(c.total_mass - 14.0) < 1e-12  # settotalmass
    # Ant torso sphere r=.25 density 5
    assert abs(a.body_mass[1] - 5 * 4 / 3 * np.pi * 0.25**3) < 1e-12
    assert (c.dof_invweight0 > 0).all() and (a.dof_invweight0 > 0).all()
    # free joint: translational / rotational dof_invweight0 are averaged
    assert np.allclose(a.dof_invweight0[:3], a.dof_invweight0[0])
    assert np.allclose(a.dof_invweight0[3:6], a.dof_invweight0[3])
    # four-fold symmetry of the Ant
    assert np.allclose(a.body_invweight0[2:5], a.body_invweight0[5:8], rtol=1e-9)


def test_free_fall_is_exact_under_rk4():
    a = RawMj("Ant")
    a.set(ANT_Q0, np.zeros(14))
    for _ in range(4):
        a.step(5)
        q, v, m = a.get()
        t = m["time"]
        assert m["ncon"] == 0 and m["nefc"] == 0
        assert abs(q[2] - (0.75 - 0.5 * 9.81 * t * t)) < 1e-12


def test_energy_conservation_rk4_and_euler_order():
    rng = np.random.default_rng(0)
    q = ANT_Q0.copy()
    q[7:] = rng.uniform(-0.3, 0.3, 8)
    q[3:7] = [0.9, 0.1, 0.3, -0.2]
    q[3:7] /= np.linalg.norm(q[3:7])
    a = RawMj("Ant", extra=FREE)
    a.set(q, rng.normal(0, 1, 14))
    e0 = sum(a.get()[2][k] for k in ("ke", "pe"))
    a.step(100)
    m = a.get()[2]
    assert abs(m["ke"] + m["pe"] - e0) < 1e-5 * abs(e0)  # O(h^4)
    assert m["asym"] == 0.0
    # cheetah: RK4 conserves, semi-implicit Euler drifts linearly in h
    qc = rng.uniform(-0.3, 0.3, 9)
    qc[1] = 0
    vc = rng.normal(0, 1, 9)
    drift = {}
    for integ, h in [(1, 0.01), (0, 0.01), (0, 0.005)]:
        c = RawMj("HalfCheetah", extra=(5, 0.1, 1, 0.1, 1, 1, 1, 1, integ, h))
        c.set(qc, vc)
        e0 = sum(c.get()[2][k] for k in ("ke", "pe"))
        c.step(int(round(0.2 / h)))
        m = c.get()[2]
        drift[(integ, h)] = m["ke"] + m["pe"] - e0
    assert abs(drift[(1, 0.01)]) < 1e-6
    assert 1.8 < drift[(0, 0.01)] / drift[(0, 0.005)] < 2.2


def test_static_equilibrium_supports_weight():
    c = RawMj("HalfCheetah")
    c.set(np.zeros(9), np.zeros(9))
    c.step(500)
    q, v, m = c.get()
    assert np.abs(v).max() < 1e-3
    assert abs(m["fsum"] - 14 * 9.81) < 1e-2  # sum of pyramid forces = m g
    assert m["fmin"] >= 0 and m["resid"] < 1e-9  # forces >= 0, KKT residual
    a = RawMj("Ant")
    a.set(ANT_Q0, np.zeros(14))
    a.step(150)
    q, v, m = a.get()
    assert m["ncon"] == 4 and abs(m["fsum"] - a.total_mass * 9.81) < 2e-3
    assert m["resid"] < 1e-9


def test_cheetah_stays_planar_and_ant_mirror_symmetry():
    a = RawMj("Ant")
    a.set(ANT_Q0, np.zeros(14))
    a.step(120)
    q, _, _ = a.get()
    # symmetric initial state, no control: x, y stay 0 and the legs stay mirrored
    assert abs(q[0]) < 1e-9 and abs(q[1]) < 1e-9
    assert np.allclose(q[8], [-q[10], -q[12], q[14]], atol=1e-9)


def test_product_planar_code_matches_oracle_on_cpu():
    """Host instantiation of the exact kernel source (mj_cheetah.hip.h; mj_ant4.hip.h with
    its lane quad emulated by Q4<double>) vs the oracle, teacher forced: two independent
    formulations."""
    from oracle.orc import Oracle

    h = os.path.join(ROOT, "tests", "cpu_harness")
    for name in ("cheetah", "ant"):
        so = os.path.join(h, f"lib{name}_host.so")
        src = os.path.join(h, f"{name}_host.cpp")
        hdrs = [os.path.join(ROOT, "envpool_amd", "csrc", f) for f in
                ("mj_cheetah.hip.h", "mj_ant.hip.h", "mj_ant4.hip.h", "mj_quad.hip.h", "mj_ant_model.h")]
        newest = max(os.path.getmtime(f) for f in [src] + hdrs)
        if not os.path.exists(so) or os.path.getmtime(so) < newest:
            subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", src, "-o", so],
                           check=True)
    rng = np.random.default_rng(3)
    for task, lib, nq, nv, nu, skip in (("HalfCheetah", "cheetah", 9, 9, 6, 1),
                                        ("Walker2d", "cheetah", 9, 9, 6, 1),
                                        ("Walker2dV5", "cheetah", 9, 9, 6, 1),
                                        ("Ant", "ant", 15, 14, 8, 2)):
        L = ctypes.CDLL(os.path.join(h, f"lib{lib}_host.so"))
        n = 8
        orc = Oracle(task, n, seed=9, max_episode_steps=1000)
        orc.reset()
        worst = 0.0
        for t in range(40):
            st = orc.get_state()
            act = rng.uniform(-1, 1, size=(n, nu))
            b = orc.step(act)
            for e in range(n):
                if b["elapsed_step"][e, 0] == 0:
                    continue
                q, v, w = st[e, :nq].copy(), st[e, nq:nq + nv].copy(), st[e, nq + nv:nq + 2 * nv].copy()
                qo, vo, wo = np.zeros(nq), np.zeros(nv), np.zeros(nv)
                it = ctypes.c_int(0)
                args = [x.ctypes.data_as(ctypes.c_void_p) for x in (q, v, w, np.ascontiguousarray(act[e]))]
                outs = [x.ctypes.data_as(ctypes.c_void_p) for x in (qo, vo, wo)]
                if lib == "ant":
                    lag = np.zeros(2)
                    L.ant_host_step(*args, 5, 0, *outs, lag.ctypes.data_as(ctypes.c_void_p), ctypes.byref(it))
                elif task.startswith("Walker2d"):
                    L.walker_host_step(*args, 4, int(task.endswith("V5")), 0, *outs,
                                       ctypes.byref(it))
                    vo = np.clip(vo, -10, 10)
                else:
                    L.cheetah_host_step(*args, 5, 0, *outs, ctypes.byref(it))
                worst = max(worst, np.abs(np.concatenate([qo[skip:], vo]) - b["obs"][e]).max())
        assert worst < 1e-9, (task, worst)


def test_product_ant_quad_layout_on_cpu():
    """mj_ant4.hip.h (one env per lane quad, Q4 emulation): the mirror structure it relies
    on holds exactly for the compiled model, fp32 instantiation stays close to fp64, and the
    Ant-v5 contact wrench (cfrc_ext of the last forward evaluation, assembled per lane and
    reduced over the quad) matches the oracle's mj_rnePostConstraint."""
    from oracle.orc import Oracle

    h = os.path.join(ROOT, "tests", "cpu_harness")
    so, src = os.path.join(h, "libant_host.so"), os.path.join(h, "ant_host.cpp")
    hdrs = [os.path.join(ROOT, "envpool_amd", "csrc", f) for f in
            ("mj_ant.hip.h", "mj_ant4.hip.h", "mj_quad.hip.h", "mj_ant_model.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in [src] + hdrs):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", src, "-o", so], check=True)
    L = ctypes.CDLL(so)
    assert L.ant_host_symmetric() == 1
    vp = ctypes.c_void_p
    n, nq, nv, nu = 8, 15, 14, 8
    # Ant-v5: use_contact_force + post_constraint, world body excluded (tests/test_gpu_mujoco.py)
    extra = (5, 0.5, 1.0, 0.1, 0, 0, 0, 0, -1, 0, 0, 3, 1, 1, 1, 0)
    orc = Oracle("Ant", n, seed=4, max_episode_steps=1000, extra=extra)
    orc.reset()
    rng = np.random.default_rng(8)
    worst, worst32, nz = 0.0, 0.0, 0
    for t in range(60):
        st = orc.get_state()
        act = rng.uniform(-1, 1, size=(n, nu))
        b = orc.step(act)
        for e in range(n):
            if b["elapsed_step"][e, 0] == 0:
                continue
            q, v, w = st[e, :nq].copy(), st[e, nq:nq + nv].copy(), st[e, nq + nv:nq + 2 * nv].copy()
            qo, vo, wo, lag, cf = np.zeros(nq), np.zeros(nv), np.zeros(nv), np.zeros(2), np.zeros(84)
            args = [x.ctypes.data_as(vp) for x in (q, v, w, np.ascontiguousarray(act[e]))]
            L.ant_host_step_wrench(*args, 5, *[x.ctypes.data_as(vp) for x in (qo, vo, wo, lag, cf)])
            obs = np.concatenate([qo[2:], vo, np.clip(cf[6:], -1.0, 1.0)])
            worst = max(worst, float(np.abs(obs - b["obs"][e]).max()))
            nz += int((np.abs(cf[6:]) > 0).sum())
            q32, v32, w32, it = np.zeros(nq), np.zeros(nv), np.zeros(nv), ctypes.c_int(0)
            L.ant_host_step(*args, 5, 1, *[x.ctypes.data_as(vp) for x in (q32, v32, w32, lag)],
                            ctypes.byref(it))
            worst32 = max(worst32, float(np.median(np.abs(np.concatenate([q32[2:], v32]) - b["obs"][e, :27]))))
    assert nz > 100, nz          # contacts were really exercised
    assert worst < 1e-9, worst
    assert worst32 < 1e-4, worst32


def test_product_pendulum_code_matches_oracle_on_cpu():
    """Host instantiation of mj_pendulum.hip.h (cart + 1 / 2 link chain, limit rows,
    RK4) vs the generic oracle, teacher forced, including states pushed against
    the slider / hinge limits."""
    from oracle.orc import Oracle

    h = os.path.join(ROOT, "tests", "cpu_harness")
    so, src = os.path.join(h, "libpendulum_host.so"), os.path.join(h, "pendulum_host.cpp")
    hdr = os.path.join(ROOT, "envpool_amd", "csrc", "mj_pendulum.hip.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", src, "-o", so], check=True)
    L = ctypes.CDLL(so)
    L.pendulum_host_step.argtypes = ([ctypes.c_int] + [ctypes.c_void_p] * 3 +
                                     [ctypes.c_double, ctypes.c_int] + [ctypes.c_void_p] * 5)
    rng = np.random.default_rng(5)
    for nl, task, amax, fs in ((1, "InvertedPendulum", 3.0, 2), (2, "InvertedDoublePendulum", 1.0, 5)):
        nv, n = nl + 1, 16
        orc = Oracle(task, n, seed=9, max_episode_steps=1000)
        orc.reset()
        worst, forced = 0.0, 0
        for t in range(40):
            st = orc.get_state()
            if t % 2:
                side = rng.choice([-1.0, 1.0], n)
                st[:, 0] = side * rng.uniform(0.97, 1.005, n)
                st[:, nv] = side * rng.uniform(0, 3, n)
                if nl == 1:
                    st[:, 1] = side * rng.uniform(1.5, 1.58, n)
                st[:, 3 * nv + 3] = 0
                orc.set_state(st)
            act = rng.uniform(-amax, amax, size=(n, 1))
            b = orc.step(act)
            for e in range(n):
                if b["elapsed_step"][e, 0] == 0:
                    continue
                q, v, w = (st[e, :nv].copy(), st[e, nv:2 * nv].copy(), st[e, 2 * nv:3 * nv].copy())
                qo, vo, wo, aux = np.zeros(nv), np.zeros(nv), np.zeros(nv), np.zeros(5)
                it = ctypes.c_int(0)
                L.pendulum_host_step(nl, q.ctypes.data, v.ctypes.data, w.ctypes.data, float(act[e, 0]), fs,
                                     qo.ctypes.data, vo.ctypes.data, wo.ctypes.data, aux.ctypes.data,
                                     ctypes.byref(it))
                if nl == 1:
                    got = np.concatenate([qo, vo])
                else:
                    got = np.concatenate([[qo[0]], np.sin(qo[1:]), np.cos(qo[1:]), np.clip(vo, -10, 10),
                                          np.clip(aux[2:5], -10, 10)])
                    forced += int(abs(b["obs"][e][8]) > 0)
                worst = max(worst, np.abs(got - b["obs"][e]).max())
        assert worst < 1e-9, (task, worst)
        assert nl == 1 or forced > 0


def test_product_reacher_code_matches_oracle_on_cpu():
    """Host instantiation of the cart-less chain (mj_pendulum.hip.h, Reacher model) vs
    the generic oracle: arm state and the lagged fingertip position."""
    from oracle.orc import Oracle

    h = os.path.join(ROOT, "tests", "cpu_harness")
    so, src = os.path.join(h, "libpendulum_host.so"), os.path.join(h, "pendulum_host.cpp")
    hdr = os.path.join(ROOT, "envpool_amd", "csrc", "mj_pendulum.hip.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", src, "-o", so], check=True)
    L = ctypes.CDLL(so)
    L.reacher_host_step.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] + [ctypes.c_void_p] * 5
    rng = np.random.default_rng(3)
    n = 16
    orc = Oracle("Reacher", n, seed=9, max_episode_steps=50)
    orc.reset()
    worst = 0.0
    for t in range(60):
        st = orc.get_state()
        if t % 4 == 3:
            st[:, 1] = rng.choice([-1.0, 1.0], n) * rng.uniform(2.9, 3.05, n)
            st[:, 5] = np.sign(st[:, 1]) * rng.uniform(0, 5, n)
            st[:, 15] = 0
            orc.set_state(st)
        act = rng.uniform(-1.2, 1.2, (n, 2))
        b = orc.step(act)
        st1 = orc.get_state()
        for e in range(n):
            if b["elapsed_step"][e, 0] == 0:
                continue
            q, v, w = st[e, 0:2].copy(), st[e, 4:6].copy(), st[e, 8:10].copy()
            qo, vo, wo, aux = np.zeros(2), np.zeros(2), np.zeros(2), np.zeros(5)
            it, a = ctypes.c_int(0), np.ascontiguousarray(act[e])
            L.reacher_host_step(q.ctypes.data, v.ctypes.data, w.ctypes.data, a.ctypes.data, 2,
                                qo.ctypes.data, vo.ctypes.data, wo.ctypes.data, aux.ctypes.data,
                                ctypes.byref(it))
            worst = max(worst, np.abs(qo - st1[e, 0:2]).max(), np.abs(vo - st1[e, 4:6]).max(),
                        abs(aux[0] - st1[e, 13]), abs(-aux[1] - st1[e, 14]))
    assert worst < 1e-10, worst


def test_product_swimmer_code_matches_oracle_on_cpu():
    """Host instantiation of the planar floating chain with the inertia-box fluid
    forces (mj_pendulum.hip.h, Swimmer model; y mirrored) vs the generic 3-D oracle."""
    from oracle.orc import Oracle

    h = os.path.join(ROOT, "tests", "cpu_harness")
    so, src = os.path.join(h, "libpendulum_host.so"), os.path.join(h, "pendulum_host.cpp")
    hdr = os.path.join(ROOT, "envpool_amd", "csrc", "mj_pendulum.hip.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", src, "-o", so], check=True)
    L = ctypes.CDLL(so)
    L.swimmer_host_step.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] + [ctypes.c_void_p] * 5
    sg = np.array([1, -1, 1, 1, 1.0])
    rng = np.random.default_rng(3)
    n = 16
    orc = Oracle("Swimmer", n, seed=9, max_episode_steps=1000)
    orc.reset()
    worst = 0.0
    for t in range(60):
        st = orc.get_state()
        if t % 5 == 4:
            st[:, 3] = rng.choice([-1.0, 1.0], n) * rng.uniform(1.70, 1.78, n)
            st[:, 8] = np.sign(st[:, 3]) * rng.uniform(0, 3, n)
            orc.set_state(st)
        act = rng.uniform(-1.2, 1.2, (n, 2))
        orc.step(act)
        st1 = orc.get_state()
        for e in range(n):
            q, v, w = sg * st[e, 0:5], sg * st[e, 5:10], sg * st[e, 10:15]
            c = np.array([0, 0, 0, act[e, 0], act[e, 1]])
            qo, vo, wo, aux = np.zeros(5), np.zeros(5), np.zeros(5), np.zeros(8)
            it = ctypes.c_int(0)
            L.swimmer_host_step(q.ctypes.data, v.ctypes.data, w.ctypes.data, c.ctypes.data, 4,
                                qo.ctypes.data, vo.ctypes.data, wo.ctypes.data, aux.ctypes.data,
                                ctypes.byref(it))
            worst = max(worst, np.abs(sg * qo - st1[e, 0:5]).max(), np.abs(sg * vo - st1[e, 5:10]).max())
    assert worst < 1e-10, worst


def test_product_hopper_code_matches_oracle_on_cpu():
    """Host instantiation of the planar tree with the Hopper model (ghost second leg,
    contact margin, capsule-capsule body pairs) vs the generic 3-D oracle, on random
    rollouts and on folded configurations where the body pairs collide."""
    from mj_util import RawMj
    from oracle.orc import Oracle

    h = os.path.join(ROOT, "tests", "cpu_harness")
    so, src = os.path.join(h, "libcheetah_host.so"), os.path.join(h, "cheetah_host.cpp")
    hdr = os.path.join(ROOT, "envpool_amd", "csrc", "mj_cheetah.hip.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", src, "-o", so], check=True)
    L = ctypes.CDLL(so)
    L.hopper_host_step.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4

    def host(q, v, w, a):
        qo, vo, wo, it = np.zeros(6), np.zeros(6), np.zeros(6), ctypes.c_int(0)
        q, v, w, a = (np.ascontiguousarray(x, dtype=np.float64) for x in (q, v, w, a))
        L.hopper_host_step(q.ctypes.data, v.ctypes.data, w.ctypes.data, a.ctypes.data, 4, 0,
                           qo.ctypes.data, vo.ctypes.data, wo.ctypes.data, ctypes.byref(it))
        assert it.value >= 0  # the ghost leg stayed at rest
        return qo, vo

    rng = np.random.default_rng(3)
    n = 8
    orc = Oracle("Hopper", n, seed=9, max_episode_steps=1000)
    orc.reset()
    worst = 0.0
    for t in range(40):
        st = orc.get_state()
        act = rng.uniform(-1.2, 1.2, (n, 3))
        b = orc.step(act)
        st1 = orc.get_state()
        for e in range(n):
            if b["elapsed_step"][e, 0] == 0:
                continue
            qo, vo = host(st[e, 0:6], st[e, 6:12], st[e, 12:18], act[e])
            worst = max(worst, np.abs(qo - st1[e, 0:6]).max(), np.abs(vo - st1[e, 6:12]).max())
    raw = RawMj("Hopper")
    noself = RawMj("Hopper", extra=[4, 1e-3, 1, 5e-3, 0, 0, 0, 0, -1, 0, 0, 3, 0, 0, 0, -1, 0, 1, 1])
    hits = 0
    for trial in range(120):
        q = np.array([0, rng.uniform(0.9, 1.4), rng.uniform(-0.5, 0.5), rng.uniform(-2.6, 0),
                      rng.uniform(-2.6, 0), rng.uniform(-0.78, 0.78)])
        v, a = rng.uniform(-2, 2, 6), rng.uniform(-1, 1, 3)
        raw.set(q, v, a)
        raw.step(4)
        noself.set(q, v, a)
        noself.step(4)
        q1, v1, _ = raw.get()
        hits += int(np.abs(q1 - noself.get()[0]).max() > 1e-9)
        qo, vo = host(q, v, np.zeros(6), a)
        worst = max(worst, np.abs(qo - q1).max(), np.abs(vo - v1).max())
    assert worst < 1e-9, worst
    assert hits >= 5  # body-body contacts were really exercised


def test_product_tree_code_matches_oracle_on_cpu():
    """Host instantiation of mj_tree.hip.h (the Humanoid / HumanoidStandup kernel source:
    static-slot constraint rows, PGS on a = qacc_smooth + M^-1 J'f, sparse L'DL) vs the
    oracle's dense generic engine, teacher forced per env-step (5 RK4 mj_steps), including
    mj_rnePostConstraint's cfrc_ext.  The standup episode lies on the floor: dozens of
    pyramidal floor contacts and frictionless self contacts per step."""
    from oracle.orc import Oracle

    csrc = os.path.join(ROOT, "envpool_amd", "csrc")
    subprocess.run(["make", "-s", "-C", csrc, "build/mj_humanoid_consts.inc"], check=True)
    h = os.path.join(ROOT, "tests", "cpu_harness")
    so, src = os.path.join(h, "libhumanoid_host.so"), os.path.join(h, "humanoid_host.cpp")
    deps = [src, os.path.join(csrc, "mj_tree.hip.h"), os.path.join(csrc, "mj_tree_model.h"),
            os.path.join(csrc, "build", "mj_humanoid_consts.inc")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", src, "-o", so], check=True)
    L = ctypes.CDLL(so)
    vp = ctypes.c_void_p
    nq, nv, nu = 24, 23, 17
    for su, task, steps in ((0, "Humanoid", 40), (1, "HumanoidStandup", 120)):
        r = RawMj(task)
        mo = np.zeros(64)
        L.humanoid_host_model(su, mo.ctypes.data_as(vp))
        assert abs(mo[0] - r.meaninertia) < 1e-12 * r.meaninertia
        np.testing.assert_allclose(mo[2:16], r.body_mass, rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(mo[16:39], r.dof_invweight0, rtol=1e-11)
        np.testing.assert_allclose(mo[39:53], r.body_invweight0[:, 0], rtol=1e-11, atol=1e-15)
        n = 8
        # extra[13] = post_constraint (v5), extra[12] = use_contact_force
        extra = [5, 0.1, 1.0 if su else 1.25, 0.01, 0, 0, 0, 0, -1, 0, 0, 3, 1, 1]
        orc = Oracle(task, n, seed=5, max_episode_steps=1000, extra=extra)
        orc.reset()
        rng = np.random.default_rng(1)
        worst, most = 0.0, 0
        for t in range(steps):
            st = orc.get_state()
            act = rng.uniform(-0.4, 0.4, size=(n, nu))
            b = orc.step(act)
            for e in range(n):
                if b["elapsed_step"][e, 0] == 0:
                    continue
                q, v, w = (st[e, :nq].copy(), st[e, nq:nq + nv].copy(),
                           st[e, nq + nv:nq + 2 * nv].copy())
                o = np.zeros(512)
                L.humanoid_host_step(q.ctypes.data_as(vp), v.ctypes.data_as(vp), w.ctypes.data_as(vp),
                                     np.ascontiguousarray(act[e]).ctypes.data_as(vp), 5, su, 1,
                                     o.ctypes.data_as(vp))
                k = nq + 2 * nv
                obs = np.concatenate([o[2:nq], o[nq:nq + nv], o[k:k + 140 + 84 + 23 + 84]])
                ref = b["obs"][e]
                worst = max(worst, float((np.abs(obs - ref) / (1.0 + np.abs(ref))).max()))
                most = max(most, int(o[k + 331 + 2]))
        print(f"{task}: worst teacher-forced rel |d obs| = {worst:.2e}; max active groups {most}")
        assert worst < 1e-8, (task, worst)
        assert most >= (10 if su else 3)


def test_product_pusher_code_matches_oracle_on_cpu():
    """Host instantiation of mj_pusher.hip.h (the Pusher kernel source: 7-dof arm with the
    fixed bodies merged into their parents, sliding cylinder, capsule/sphere-vs-cylinder and
    table-plane contacts) vs the oracle's generic engine on the full 13-body model: compiled
    model constants, then teacher-forced env-steps (5 mj_steps) from states that put the
    cylinder within reach of the fingertips so the contact branches run."""
    from oracle.orc import Oracle

    csrc = os.path.join(ROOT, "envpool_amd", "csrc")
    h = os.path.join(ROOT, "tests", "cpu_harness")
    so, src = os.path.join(h, "libpusher_host.so"), os.path.join(h, "pusher_host.cpp")
    deps = [src, os.path.join(csrc, "mj_pusher.hip.h"), os.path.join(csrc, "mj_pusher_model.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", src, "-o", so], check=True)
    L = ctypes.CDLL(so)
    vp = ctypes.c_void_p

    def host(q, v, w, a, nsub, v5):
        qo, vo, wo, lag = np.zeros(9), np.zeros(9), np.zeros(9), np.zeros(5)
        it = ctypes.c_int(0)
        L.pusher_host_step(*[vp(x.ctypes.data) for x in (q, v, w, a)], nsub, v5,
                           *[vp(x.ctypes.data) for x in (qo, vo, wo, lag)], ctypes.byref(it))
        return qo, vo, wo, lag

    lo = np.array([-2.2854, -0.5236, -1.5, -2.3213, -1.5, -1.094, -1.5])
    hi = np.array([1.714602, 1.3963, 1.7, 0.0, 1.5, 0.0, 1.5])
    n, nq, nv = 16, 11, 11
    for task, v5 in (("Pusher", 0), ("PusherV5", 1)):
        r = RawMj(task)
        mo = np.zeros(32)
        L.pusher_host_model(v5, vp(mo.ctypes.data))
        bm = r.body_mass
        # fixed bodies merge into their parents: links = bodies [1, 2, 3+4, 5, 6+7, 8, 9+10]
        want = [bm[1], bm[2], bm[3] + bm[4], bm[5], bm[6] + bm[7], bm[8], bm[9] + bm[10]]
        np.testing.assert_allclose(mo[:7], want, rtol=1e-13)
        np.testing.assert_allclose(mo[7:14], r.dof_invweight0[:7], rtol=1e-11)
        np.testing.assert_allclose(mo[14:17], [r.body_invweight0[9, 0], r.body_invweight0[11, 0], bm[11]],
                                   rtol=1e-11)
        orc = Oracle(task, n, seed=9, max_episode_steps=1000)
        orc.reset()
        rng = np.random.default_rng(5)
        worst = 0.0
        ncyl = nplane = 0
        buf = np.zeros(640)
        orc.lib.mjcpu_raw_contacts.restype = ctypes.c_int
        orc.lib.mjcpu_raw_contacts.argtypes = [vp, ctypes.c_int, vp]
        inner = ctypes.cast(orc.h, ctypes.POINTER(_H)).contents.h
        for t in range(12):
            orc.step(rng.uniform(-2, 2, (n, 7)))
            st = orc.get_state()
            for e in range(n):
                q = rng.uniform(lo * 0.6, hi * 0.6)
                q[1], q[3] = rng.uniform(0.25, 0.75), rng.uniform(-0.8, 0.0)
                v = rng.normal(0, 0.5, 7)
                z2 = np.zeros(2)
                lag = host(np.concatenate([q, z2]), np.concatenate([v, z2]), np.zeros(9),
                           np.zeros(7), 1, v5)[3]    # fingertip position at ~q
                ang, dist = rng.uniform(0, 2 * np.pi), rng.uniform(0.0, 0.2)
                ox, oy = lag[0] + dist * np.cos(ang), lag[1] + dist * np.sin(ang)
                st[e, :7], st[e, 7], st[e, 8] = q, oy + 0.05, ox - 0.45
                st[e, nq:nq + 7], st[e, nq + 7:nq + 9] = v, rng.normal(0, 0.05, 2)
                st[e, nq + nv:nq + 2 * nv] = 0
                st[e, -5:-2], st[e, -2], st[e, -1] = lag[:3], ox, oy
            orc.set_state(st)
            act = rng.uniform(-2, 2, (n, 7))
            b = orc.step(act)
            st2 = orc.get_state()
            for e in range(n):
                if b["elapsed_step"][e, 0] == 0:
                    continue
                qo, vo, wo, lag = host(st[e, :9].copy(), st[e, nq:nq + 9].copy(),
                                       st[e, nq + nv:nq + nv + 9].copy(),
                                       np.ascontiguousarray(act[e]), 5, v5)
                ob = np.concatenate([qo[:7], vo[:7], lag])
                worst = max(worst, np.abs(ob - b["obs"][e, :19]).max(),
                            np.abs(qo[7:9] - st2[e, 7:9]).max(),
                            np.abs(vo[7:9] - st2[e, nq + 7:nq + 9]).max())
                nc = orc.lib.mjcpu_raw_contacts(inner, e, vp(buf.ctypes.data))
                for c in range(nc):
                    if buf[10 * c + 9] >= 0:    # active row
                        ncyl += int(buf[10 * c] != 0)
                        nplane += int(buf[10 * c] == 0)
        assert worst < 1e-9, (task, worst)
        assert ncyl > 20 and nplane > 20, (task, ncyl, nplane)


def test_product_humanoid_quad_code_matches_oracle_on_cpu():
    """Host instantiation of mj_hum4.hip.h -- the Humanoid / HumanoidStandup kernel source with
    one env split over a lane quad (trunk replicated, one limb per lane, arrow-structured L'DL,
    rows kept as y = L^-T J', y-space PGS), the quad emulated by Q4<double> -- vs the oracle's
    dense generic engine, teacher forced per env-step (5 RK4 mj_steps) incl. cfrc_ext; and the
    smooth dynamics (qacc_smooth, cinert, cvel, geoms) vs the one-env-per-lane mj_tree.hip.h."""
    from oracle.orc import Oracle

    csrc = os.path.join(ROOT, "envpool_amd", "csrc")
    subprocess.run(["make", "-s", "-C", csrc, "build/mj_humanoid_consts.inc"], check=True)
    h = os.path.join(ROOT, "tests", "cpu_harness")
    so, src = os.path.join(h, "libhumanoid4_host.so"), os.path.join(h, "humanoid4_host.cpp")
    deps = [src] + [os.path.join(csrc, f) for f in ("mj_hum4.hip.h", "mj_quad.hip.h", "mj_tree.hip.h",
                                                     "mj_tree_model.h", "build/mj_humanoid_consts.inc")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", src, "-o", so], check=True)
    L = ctypes.CDLL(so)
    vp = ctypes.c_void_p
    nq, nv, nu = 24, 23, 17
    rng = np.random.default_rng(0)
    for su in (0, 1):
        worst = 0.0
        for _ in range(20):
            q = np.zeros(nq)
            q[:2], q[2], q[3:7] = rng.normal(0, 1, 2), (0.1 if su else 1.4), rng.normal(0, 1, 4)
            q[7:] = rng.uniform(-0.8, 0.8, 17)
            v, ctrl = rng.normal(0, 1.5, nv), rng.uniform(-0.6, 0.6, nu)
            a, b = np.zeros(381), np.zeros(381)
            for use_tree, o in ((0, a), (1, b)):
                L.hum4_smooth(vp(q.ctypes.data), vp(v.ctypes.data), vp(ctrl.ctypes.data), su, use_tree,
                              vp(o.ctypes.data))
            a[273:].reshape(18, 6)[[2, 8, 11, 14, 17], 3:] = 0  # sphere "axes": unused
            worst = max(worst, float((np.abs(a - b) / (1.0 + np.abs(b))).max()))
        assert worst < 1e-10, (su, worst)
    # variants 2 / 3: the same models with 4 / 8 register rows (+ 8 / 12 overflow rows), so that nearly every
    # solve takes the hybrid form of the PGS (tracked register rows + on-chip overflow rows; the product holds
    # 12 + 8 / 16 + 16) and some fall through to the streaming form
    for su, task, steps in ((0, "Humanoid", 40), (1, "HumanoidStandup", 120), (2, "Humanoid", 30),
                            (3, "HumanoidStandup", 60)):
        n = 8
        extra = [5, 0.1, 1.0 if su & 1 else 1.25, 0.01, 0, 0, 0, 0, -1, 0, 0, 3, 1, 1]
        orc = Oracle(task, n, seed=5, max_episode_steps=1000, extra=extra)
        orc.reset()
        rng = np.random.default_rng(1)
        worst, most, hybrid, stream, rows, again = 0.0, 0, 0, 0, 0, 0
        for t in range(steps):
            st = orc.get_state()
            act = rng.uniform(-0.4, 0.4, size=(n, nu))
            b = orc.step(act)
            for e in range(n):
                if b["elapsed_step"][e, 0] == 0:
                    continue
                q, v, w = (st[e, :nq].copy(), st[e, nq:nq + nv].copy(),
                           st[e, nq + nv:nq + 2 * nv].copy())
                o = np.zeros(512)
                L.humanoid4_host_step(vp(q.ctypes.data), vp(v.ctypes.data), vp(w.ctypes.data),
                                      vp(np.ascontiguousarray(act[e]).ctypes.data), 5, su, 1,
                                      vp(o.ctypes.data))
                k = nq + 2 * nv
                obs = np.concatenate([o[2:nq], o[nq:nq + nv], o[k:k + 140 + 84 + 23 + 84]])
                ref = b["obs"][e]
                worst = max(worst, float((np.abs(obs - ref) / (1.0 + np.abs(ref))).max()))
                most = max(most, int(o[k + 331 + 2]))
                hybrid += int(o[k + 331 + 3]) % 1000
                stream += int(o[k + 331 + 3]) // 1000
                rows = max(rows, int(o[k + 331 + 4]))
                again += int(o[k + 331 + 5])
        print(f"{task} (quad layout, variant {su}): worst teacher-forced rel |d obs| = {worst:.2e}; max active "
              f"groups {most}, most rows {rows}, hybrid solves {hybrid}, streaming solves {stream}, started over {again}")
        assert worst < 1e-8, (task, worst)
        assert most >= (10 if su & 1 else 3)
        if su >= 2:
            assert hybrid > 100 * steps // 30, (task, hybrid)


def test_product_planar_lane_group_code_matches_oracle_on_cpu():
    """Host instantiation of the lane-group planar step (mj_planar_lg.hip.h: one env over 2 or 4 lanes,
    emulated by LV<double, KL>) vs the oracle and vs the one-env-per-lane formulation
    (mj_cheetah.hip.h), teacher forced.  The harness also checks that every value that must be
    replicated over the group (torso state, the parity lanes of a leg) is bit-identical in all lanes."""
    from oracle.orc import Oracle

    h = os.path.join(ROOT, "tests", "cpu_harness")
    csrc = os.path.join(ROOT, "envpool_amd", "csrc")
    for name, hdrs in (("planar_lg", ("mj_planar_lg.hip.h", "mj_cheetah.hip.h", "mj_cheetah_model.h")),
                       ("cheetah", ("mj_cheetah.hip.h", "mj_cheetah_model.h"))):
        so, src = os.path.join(h, f"lib{name}_host.so"), os.path.join(h, f"{name}_host.cpp")
        newest = max(os.path.getmtime(f) for f in [src] + [os.path.join(csrc, x) for x in hdrs])
        if not os.path.exists(so) or os.path.getmtime(so) < newest:
            subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", src, "-o", so], check=True)
    # the same source with the exact line search from the SECOND Newton trip of a forward pass on (the product: one
    # evaluation per trip, exact search only as a fallback from trip 8 that the benchmark never reaches): the fallback's
    # code path, and that both searches end on the oracle's minimiser
    so2, src2 = os.path.join(h, "libplanar_lg_host_exact.so"), os.path.join(h, "planar_lg_host.cpp")
    newest = max(os.path.getmtime(f) for f in [src2] + [os.path.join(csrc, x) for x in
                                                       ("mj_planar_lg.hip.h", "mj_cheetah.hip.h", "mj_cheetah_model.h")])
    if not os.path.exists(so2) or os.path.getmtime(so2) < newest:
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-DEPA_LG_LS_EXACT_AFTER=1", src2, "-o", so2],
                       check=True)
    Lo = ctypes.CDLL(os.path.join(h, "libcheetah_host.so"))
    vp = ctypes.c_void_p
    for L in (ctypes.CDLL(os.path.join(h, "libplanar_lg_host.so")), ctypes.CDLL(so2)):
        _planar_lane_group_vs_oracle(L, Lo)


def _planar_lane_group_vs_oracle(L, Lo):
    from oracle.orc import Oracle

    vp = ctypes.c_void_p
    L.planar_lg_step.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, ctypes.c_int, vp, vp, vp, vp]
    rng = np.random.default_rng(3)
    for task, model, nsub in (("HalfCheetah", 0, 5), ("Walker2d", 1, 4), ("Walker2dV5", 2, 4), ("Hopper", 3, 4)):
        for kl in ((1,) if model == 3 else (2, 4)):  # the one-legged Hopper: a group of ONE lane
            n = 8
            orc = Oracle(task, n, seed=9, max_episode_steps=1000)
            orc.reset()
            worst, worst_lane = 0.0, 0.0
            for t in range(40):
                st = orc.get_state()
                act = rng.uniform(-1, 1, size=(n, 3 if model == 3 else 6))
                b = orc.step(act)
                for e in range(n):
                    if b["elapsed_step"][e, 0] == 0:
                        continue
                    nd = 6 if model == 3 else 9  # the Hopper owns dofs 0..5 / motors 0..2 of the 9-dof tree
                    q, v, w, a = np.zeros(9), np.zeros(9), np.zeros(9), np.zeros(6)
                    q[:nd], v[:nd], w[:nd] = st[e, :nd], st[e, nd:2 * nd], st[e, 2 * nd:3 * nd]
                    a[:nd - 3] = act[e]
                    qo, vo, wo, it = np.zeros(9), np.zeros(9), np.zeros(9), ctypes.c_int(0)
                    rc = L.planar_lg_step(model, kl, q.ctypes.data, v.ctypes.data, w.ctypes.data, a.ctypes.data,
                                          nsub, qo.ctypes.data, vo.ctypes.data, wo.ctypes.data, ctypes.byref(it))
                    assert rc == 0, (task, kl, rc)  # replicated values agree bit for bit
                    vv = np.clip(vo, -10, 10) if model else vo
                    worst = max(worst, np.abs(np.concatenate([qo[1:nd], vv[:nd]]) - b["obs"][e]).max())
                    q2, v2, w2, it2 = np.zeros(9), np.zeros(9), np.zeros(9), ctypes.c_int(0)
                    args = [x.ctypes.data_as(vp) for x in (q, v, w, a)]
                    outs = [x.ctypes.data_as(vp) for x in (q2, v2, w2)]
                    if model == 3:
                        Lo.hopper_host_step(*args, 4, 0, *outs, ctypes.byref(it2))
                    elif model:
                        Lo.walker_host_step(*args, 4, int(model == 2), 0, *outs, ctypes.byref(it2))
                    else:
                        Lo.cheetah_host_step(*args, 5, 0, *outs, ctypes.byref(it2))
                    worst_lane = max(worst_lane, np.abs(qo - q2).max(), np.abs(vo - v2).max())
            assert worst < 1e-9 and worst_lane < 1e-9, (task, kl, worst, worst_lane)


def test_hopper_lane_group_body_pair_rows_on_cpu():
    """The Hopper's capsule-capsule body pairs (torso-leg, torso-foot, thigh-foot) in the lane-group form (a group of
    ONE lane, mj_planar_lg.hip.h) on FOLDED poses, where they touch: host instantiation vs the one-env-per-lane form
    (mj_cheetah.hip.h; must agree to rounding on every state, also the degenerate ones) and vs the oracle (the states
    with deeply crossed capsule axes, where the contact normal is numerical noise in every implementation, show up
    as rare outliers and are bounded in number, as in the GPU test)."""
    from oracle.orc import Oracle

    h = os.path.join(ROOT, "tests", "cpu_harness")
    csrc = os.path.join(ROOT, "envpool_amd", "csrc")
    for name, hdrs in (("planar_lg", ("mj_planar_lg.hip.h", "mj_cheetah.hip.h", "mj_cheetah_model.h")),
                       ("cheetah", ("mj_cheetah.hip.h", "mj_cheetah_model.h"))):
        so, src = os.path.join(h, f"lib{name}_host.so"), os.path.join(h, f"{name}_host.cpp")
        newest = max(os.path.getmtime(f) for f in [src] + [os.path.join(csrc, x) for x in hdrs])
        if not os.path.exists(so) or os.path.getmtime(so) < newest:
            subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", src, "-o", so], check=True)
    L = ctypes.CDLL(os.path.join(h, "libplanar_lg_host.so"))
    Lo = ctypes.CDLL(os.path.join(h, "libcheetah_host.so"))
    vp = ctypes.c_void_p
    L.planar_lg_step.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, ctypes.c_int, vp, vp, vp, vp]
    n = 16
    rng = np.random.default_rng(3)
    orc = Oracle("Hopper", n, seed=9, max_episode_steps=1000)
    noself = Oracle("Hopper", n, seed=9, max_episode_steps=1000,
                    extra=(4, 1e-3, 1, 5e-3, 0, 0, 0, 0, -1, 0, 0, 3, 0, 0, 0, -1, 0, 1, 1))
    orc.reset(), noself.reset()
    worst_lane, pair_hits, compared, outliers = 0.0, 0, 0, 0
    for t in range(80):
        st = orc.get_state()
        if t % 4 == 3:  # fold the leg: thigh / knee deep into their range, random height
            st[:, 1] = rng.uniform(0.9, 1.4, n)
            st[:, 2] = rng.uniform(-0.5, 0.5, n)
            st[:, 3] = rng.uniform(-2.6, 0, n)
            st[:, 4] = rng.uniform(-2.6, 0, n)
            st[:, 5] = rng.uniform(-0.78, 0.78, n)
            st[:, 6:12] = rng.uniform(-2, 2, (n, 6))
            st[:, 12:18] = 0
            st[:, 21] = 0
            orc.set_state(st)
        noself.set_state(st)
        act = rng.uniform(-1.2, 1.2, size=(n, 3))
        b, c = orc.step(act), noself.step(act)
        for e in range(n):
            if b["elapsed_step"][e, 0] == 0:
                continue
            q, v, w, a = np.zeros(9), np.zeros(9), np.zeros(9), np.zeros(6)
            q[:6], v[:6], w[:6], a[:3] = st[e, :6], st[e, 6:12], st[e, 12:18], act[e]
            qo, vo, wo, it = np.zeros(9), np.zeros(9), np.zeros(9), ctypes.c_int(0)
            rc = L.planar_lg_step(3, 1, q.ctypes.data, v.ctypes.data, w.ctypes.data, a.ctypes.data, 4,
                                  qo.ctypes.data, vo.ctypes.data, wo.ctypes.data, ctypes.byref(it))
            assert rc == 0
            q2, v2, w2, it2 = np.zeros(9), np.zeros(9), np.zeros(9), ctypes.c_int(0)
            Lo.hopper_host_step(q.ctypes.data_as(vp), v.ctypes.data_as(vp), w.ctypes.data_as(vp), a.ctypes.data_as(vp),
                                4, 0, q2.ctypes.data_as(vp), v2.ctypes.data_as(vp), w2.ctypes.data_as(vp),
                                ctypes.byref(it2))
            worst_lane = max(worst_lane, np.abs(qo - q2).max(), np.abs(vo - v2).max())
            obs = np.concatenate([qo[1:6], np.clip(vo[:6], -10, 10)])
            compared += 1
            outliers += bool(np.abs(obs - b["obs"][e]).max() > 1e-9)
            pair_hits += bool(np.abs(b["obs"][e] - c["obs"][e]).max() > 1e-9)
    assert worst_lane < 1e-9, worst_lane           # the two product forms agree everywhere
    assert pair_hits > 50, pair_hits               # the body-pair rows were exercised
    assert outliers <= compared // 200, (outliers, compared)  # vs the oracle: crossed-axes states only


def test_pusher_capsule_cylinder_rule_in_the_degenerate_poses():
    """Pusher's wrist capsule vs the object's cylinder goes through MuJoCo's convex collider; product
    (mj_pusher.hip.h::CapsuleCylinder) and oracle (oracle/mjcpu/engine.c::capcyl_contact) restate it as the
    closest points between the capsule's axis segment and the solid cylinder, with the MIDPOINT of the
    closest stretch when the minimiser is not unique.  The two independently written routines must agree
    everywhere -- in general position and exactly in the tie cases (axis parallel to the cylinder's side,
    to a flat face, along its axis, inside the solid) -- and the tie cases must obey the stated rule.
    PARITY NOTE: this pins product == oracle only.  How MuJoCo's GJK / EPA resolves a non-unique closest
    pair (and its 1e-6 tolerance) is not pinned by anything here (no MuJoCo wheel)."""
    import ctypes

    from oracle import orc

    h = os.path.join(ROOT, "tests", "cpu_harness")
    csrc = os.path.join(ROOT, "envpool_amd", "csrc")
    so, src = os.path.join(h, "libpusher_host.so"), os.path.join(h, "pusher_host.cpp")
    deps = [src, os.path.join(csrc, "mj_pusher.hip.h"), os.path.join(csrc, "mj_pusher_model.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", src, "-o", so], check=True)
    P = ctypes.CDLL(so)
    O = ctypes.CDLL(orc.PORT_LIB)
    vp, dbl = ctypes.c_void_p, ctypes.c_double
    for f in (P.pusher_host_capcyl, O.mjcpu_capsule_cylinder):
        f.argtypes = [vp, vp, dbl, vp, dbl, dbl, vp]
        f.restype = None

    def both(p0, p1, rc, c, R, H):
        outs = []
        for f in (P.pusher_host_capcyl, O.mjcpu_capsule_cylinder):
            a, b, cc, o = (np.ascontiguousarray(x, dtype=np.float64) for x in (p0, p1, c, np.zeros(7)))
            f(a.ctypes.data, b.ctypes.data, rc, cc.ctypes.data, R, H, o.ctypes.data)
            outs.append(o)
        np.testing.assert_allclose(outs[0], outs[1], rtol=0, atol=1e-12, err_msg=f"{p0} {p1}")
        return outs[0]

    R, H, rc, c = 0.05, 0.05, 0.02, np.array([0.45, -0.05, -0.275])
    rng = np.random.default_rng(0)
    for _ in range(2000):  # general position, all regions (beside, above, diagonal, penetrating)
        p0 = c + rng.uniform(-0.2, 0.2, 3)
        p1 = p0 + rng.uniform(-0.12, 0.12, 3)
        o = both(p0, p1, rc, c, R, H)
        assert abs(np.linalg.norm(o[4:7]) - 1) < 1e-12
    # axis parallel to the cylinder's axis, beside it, overlapping in z over [-0.03, 0.05]:
    # every point of the overlap is closest -> its midpoint z = 0.01, normal radial
    o = both(c + [0.1, 0, -0.03], c + [0.1, 0, 0.09], rc, c, R, H)
    np.testing.assert_allclose(o[0], 0.1 - R - rc, atol=1e-12)
    np.testing.assert_allclose(o[4:7], [-1, 0, 0], atol=1e-12)
    np.testing.assert_allclose(o[3] - c[2], 0.01, atol=1e-9)
    # axis parallel to the top face, above it, crossing the whole disc: midpoint of the chord over the disc
    o = both(c + [-0.2, 0.01, 0.09], c + [0.2, 0.01, 0.09], rc, c, R, H)
    np.testing.assert_allclose(o[0], 0.04 - rc, atol=1e-12)
    np.testing.assert_allclose(o[4:7], [0, 0, -1], atol=1e-12)
    np.testing.assert_allclose(o[1:3] - c[:2], [0.0, 0.01], atol=1e-9)
    # along the cylinder's own axis, above it: the lower end is closest (unique)
    o = both(c + [0, 0, 0.08], c + [0, 0, 0.2], rc, c, R, H)
    np.testing.assert_allclose(o[0], 0.03 - rc, atol=1e-12)
    # axis inside the solid: pushed out radially, surfaces "distance" = -rc
    o = both(c + [0.01, 0, -0.01], c + [0.02, 0, 0.01], rc, c, R, H)
    np.testing.assert_allclose(o[0], -rc, atol=1e-12)
    np.testing.assert_allclose(o[4:7], [-1, 0, 0], atol=1e-9)


def test_lane_group_solver_on_adversarial_states():
    """The one-evaluation line search of the lane-group Newton solver (mj_planar_lg.hip.h::Solve; exact search only
    as a fallback from trip 8) far from the benchmark's states: deep penetrations, joints beyond their ranges,
    velocities ~ N(0, 8), a garbage warm start ~ N(0, 50).  One env-step from each such state, host instantiation of
    the kernel source vs the oracle (whose solver searches its lines exactly): same result to rounding, i.e. the
    solver still ends on the minimiser, and the trip counts stay far from the cap of 50 per forward pass."""
    from oracle.orc import Oracle

    h = os.path.join(ROOT, "tests", "cpu_harness")
    csrc = os.path.join(ROOT, "envpool_amd", "csrc")
    so, src = os.path.join(h, "libplanar_lg_host.so"), os.path.join(h, "planar_lg_host.cpp")
    newest = max(os.path.getmtime(f) for f in [src] + [os.path.join(csrc, x) for x in
                                                      ("mj_planar_lg.hip.h", "mj_cheetah.hip.h", "mj_cheetah_model.h")])
    if not os.path.exists(so) or os.path.getmtime(so) < newest:
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", src, "-o", so], check=True)
    L = ctypes.CDLL(so)
    vp = ctypes.c_void_p
    L.planar_lg_step.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, ctypes.c_int, vp, vp, vp, vp]
    rng = np.random.default_rng(11)
    for task, model, nsub, nd, passes in (("HalfCheetah", 0, 5, 9, 5), ("Walker2d", 1, 4, 9, 16), ("Hopper", 3, 4, 6, 16)):
        n = 96
        orc = Oracle(task, n, seed=5, max_episode_steps=1000)
        orc.reset()
        worst, its = 0.0, []
        for rep in range(3):
            st = orc.get_state()
            st[:, 1] = rng.uniform(-0.3, 0.6, n) if model == 0 else rng.uniform(0.3, 1.5, n)
            st[:, 2] = rng.uniform(-3, 3, n)
            st[:, 3:nd] = rng.uniform(-1.5, 1.5, (n, nd - 3))
            st[:, nd:2 * nd] = rng.normal(0, 8, (n, nd))
            st[:, 2 * nd:3 * nd] = rng.normal(0, 50, (n, nd))
            orc.set_state(st)
            act = rng.uniform(-1, 1, (n, nd - 3))
            b = orc.step(act)
            for e in range(n):
                if b["elapsed_step"][e, 0] == 0:
                    continue
                q, v, w, a = np.zeros(9), np.zeros(9), np.zeros(9), np.zeros(6)
                q[:nd], v[:nd], w[:nd] = st[e, :nd], st[e, nd:2 * nd], st[e, 2 * nd:3 * nd]
                a[:nd - 3] = act[e]
                qo, vo, wo, it = np.zeros(9), np.zeros(9), np.zeros(9), ctypes.c_int(0)
                rc = L.planar_lg_step(model, 1 if model == 3 else 2, q.ctypes.data, v.ctypes.data, w.ctypes.data,
                                      a.ctypes.data, nsub, qo.ctypes.data, vo.ctypes.data, wo.ctypes.data, ctypes.byref(it))
                assert rc == 0
                vv = np.clip(vo, -10, 10) if model else vo
                ref, got = b["obs"][e], np.concatenate([qo[1:nd], vv[:nd]])
                worst = max(worst, (np.abs(got - ref) / (1 + np.abs(ref))).max())
                its.append(it.value)
        assert len(its) > n and worst < 1e-9, (task, worst)
        assert max(its) < 8 * passes, (task, max(its))  # no forward pass anywhere near the cap


def test_ant_solver_on_adversarial_states():
    """The Ant quad kernel's Newton solver (mj_ant4.hip.h: one line-search evaluation per trip, exact search as a
    fallback from trip 8) far from the benchmark's states: random torso orientation and height (legs deep in the floor ..
    airborne), joints anywhere, velocities ~ N(0, 4), a garbage warm start ~ N(0, 30).  One env-step from each state,
    host instantiation (lane quad emulated by Q4<double>) vs the oracle, whose solver searches exactly."""
    from oracle.orc import Oracle

    h = os.path.join(ROOT, "tests", "cpu_harness")
    so, src = os.path.join(h, "libant_host.so"), os.path.join(h, "ant_host.cpp")
    hdrs = [os.path.join(ROOT, "envpool_amd", "csrc", f) for f in
            ("mj_ant.hip.h", "mj_ant4.hip.h", "mj_quad.hip.h", "mj_ant_model.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in [src] + hdrs):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", src, "-o", so], check=True)
    L = ctypes.CDLL(so)
    rng = np.random.default_rng(12)
    n, nq, nv, nu = 64, 15, 14, 8
    orc = Oracle("Ant", n, seed=5, max_episode_steps=1000)
    orc.reset()
    worst, its = 0.0, []
    for rep in range(3):
        st = orc.get_state()
        st[:, 2] = rng.uniform(0.15, 0.9, n)
        quat = rng.normal(0, 1, (n, 4))
        st[:, 3:7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
        st[:, 7:15] = rng.uniform(-1.2, 1.2, (n, 8))
        st[:, nq:nq + nv] = rng.normal(0, 4, (n, nv))
        st[:, nq + nv:nq + 2 * nv] = rng.normal(0, 30, (n, nv))
        orc.set_state(st)
        act = rng.uniform(-1, 1, (n, nu))
        b = orc.step(act)
        for e in range(n):
            if b["elapsed_step"][e, 0] == 0 or not np.isfinite(b["obs"][e]).all():
                continue
            q, v, w = st[e, :nq].copy(), st[e, nq:nq + nv].copy(), st[e, nq + nv:nq + 2 * nv].copy()
            qo, vo, wo, it, lag = np.zeros(nq), np.zeros(nv), np.zeros(nv), ctypes.c_int(0), np.zeros(2)
            args = [x.ctypes.data_as(ctypes.c_void_p) for x in (q, v, w, np.ascontiguousarray(act[e]))]
            outs = [x.ctypes.data_as(ctypes.c_void_p) for x in (qo, vo, wo)]
            L.ant_host_step(*args, 5, 0, *outs, lag.ctypes.data_as(ctypes.c_void_p), ctypes.byref(it))
            ref, got = b["obs"][e], np.concatenate([qo[2:], vo])
            worst = max(worst, (np.abs(got - ref) / (1 + np.abs(ref))).max())
            its.append(it.value)
    assert len(its) > n and worst < 1e-9, worst
    assert max(its) < 8 * 20, max(its)  # 20 forward passes per env-step: none anywhere near the cap of 50 trips
