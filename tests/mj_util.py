"""Raw access to oracle/mjcpu for invariant tests (test infrastructure)."""
import ctypes

import numpy as np

from oracle.orc import Oracle


class _H(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("h", ctypes.c_void_p), ("n", ctypes.c_int)]


class RawMj:
    """extra = [frame_skip, ctrl_w, fwd_w, noise, no_contact, no_limit, no_act,
    no_passive]"""

    def __init__(self, task, extra=()):
        self.o = Oracle(task, 1, seed=0, max_episode_steps=1000, extra=extra)
        self.L = self.o.lib
        self.inner = ctypes.cast(self.o.h, ctypes.POINTER(_H)).contents.h
        sc = np.zeros(256)
        self.L.mjcpu_model_scalars.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.L.mjcpu_model_scalars(self.inner, sc.ctypes.data)
        self.nq, self.nv, self.nu, self.nbody, self.ngeom = (int(x) for x in sc[:5])
        self.meaninertia, self.total_mass = sc[5], sc[6]
        k = 7
        self.body_mass = sc[k:k + self.nbody].copy(); k += self.nbody
        self.dof_invweight0 = sc[k:k + self.nv].copy(); k += self.nv
        self.body_invweight0 = sc[k:k + 2 * self.nbody].reshape(-1, 2).copy()
        vp = ctypes.c_void_p
        self.L.mjcpu_raw_set.argtypes = [vp, ctypes.c_int, vp, vp, vp]
        self.L.mjcpu_raw_step.argtypes = [vp, ctypes.c_int, ctypes.c_int]
        self.L.mjcpu_raw_get.argtypes = [vp, ctypes.c_int, vp, vp, vp]

    def set(self, qpos, qvel, ctrl=None):
        qpos = np.ascontiguousarray(qpos, dtype=np.float64)
        qvel = np.ascontiguousarray(qvel, dtype=np.float64)
        ctrl = np.zeros(self.nu) if ctrl is None else np.ascontiguousarray(ctrl, dtype=np.float64)
        self.L.mjcpu_raw_set(self.inner, 0, qpos.ctypes.data, qvel.ctypes.data, ctrl.ctypes.data)

    def set_warm(self, qpos, qvel, ctrl, warm):
        """State of a running simulation incl. qacc_warmstart, no forward pass."""
        arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (qpos, qvel, ctrl, warm)]
        self.L.mjcpu_raw_set_warm.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4
        self.L.mjcpu_raw_set_warm(self.inner, 0, *[a.ctypes.data for a in arrs])

    def observed(self):
        """cinert, cvel, qfrc_actuator, cfrc_ext of the last forward evaluation."""
        nb, nv = self.nbody, self.nv
        out = np.zeros(nb * 22 + nv)
        self.L.mjcpu_raw_observed.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        self.L.mjcpu_raw_observed(self.inner, 0, out.ctypes.data)
        return (out[:nb * 10].reshape(nb, 10), out[nb * 10:nb * 16].reshape(nb, 6),
                out[nb * 16:nb * 16 + nv], out[nb * 16 + nv:].reshape(nb, 6))

    def step(self, n=1):
        self.L.mjcpu_raw_step(self.inner, 0, n)

    def get(self):
        qpos, qvel, misc = np.zeros(self.nq), np.zeros(self.nv), np.zeros(16)
        self.L.mjcpu_raw_get(self.inner, 0, qpos.ctypes.data, qvel.ctypes.data, misc.ctypes.data)
        names = ["ke", "pe", "ncon", "nefc", "iters", "time", "fmin", "resid",
                 "asym", "torso_z", "fsum"]
        return qpos, qvel, dict(zip(names, misc))
