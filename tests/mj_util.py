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


# ---- positional `extra` of mjcpu_create / ref_mujoco_driver.cc (oracle/mjcpu/tasks.c) --------
_EXTRA_NAMES = {
    "frame_skip": 0, "ctrl_cost_weight": 1, "forward_reward_weight": 2, "reset_noise_scale": 3,
    "disable_contact": 4, "disable_limit": 5, "disable_actuation": 6, "disable_passive": 7,
    "integrator": 8, "timestep": 9, "reward_if_not_terminated": 10, "constraint_obs_dim": 11,
    "use_contact_force": 12, "post_constraint": 13, "exclude_worldbody": 14,
    "legacy_healthy_reward": 15, "reward_after_step": 16, "obs_include_z": 17,
    "disable_selfcollide": 18, "exclude_root_actuator": 19, "dist_cost_weight": 20,
    "near_cost_weight": 21, "weighted_reward_info": 22, "frame_stack": 23, "warmstart_rule": 24,
}


def mj_extra(task, **over):
    """Full `extra` tuple for `task` with the task's own defaults and `over` applied."""
    base = task.replace("V5", "")
    fs = 4 if base in ("Walker2d", "Swimmer", "Hopper") else 2 if base in ("InvertedPendulum", "Reacher") else 5
    cw = {"Ant": 0.5, "Walker2d": 1e-3, "Hopper": 1e-3, "Reacher": 1.0, "Swimmer": 1e-4}.get(base, 0.1)
    fw = 1.25 if base == "Humanoid" else 1.0
    noise = 5e-3 if base in ("Walker2d", "Hopper") else 1e-2 if base in ("InvertedPendulum", "Humanoid", "HumanoidStandup") else 0.1
    ex = [fs, cw, fw, noise, 0, 0, 0, 0, -1, 0, 0, 3, 0, 0, 0, -1, 0, 1, 0, 0, 1.0, 0.5, 0, 1, 0]
    for k, v in over.items():
        ex[_EXTRA_NAMES[k]] = float(v)
    return tuple(float(v) for v in ex)


# registered gym-MuJoCo ids -> (oracle task, max_episode_steps, options); the options are those
# envpool/mujoco/gym/registration.py:38-93 passes for the version
GYM_VARIANTS = {
    "HalfCheetah-v4": ("HalfCheetah", 1000, {}),
    "HalfCheetah-v5": ("HalfCheetah", 1000, dict(post_constraint=1)),
    "Ant-v3": ("Ant", 1000, dict(use_contact_force=1)),
    "Ant-v4": ("Ant", 1000, {}),
    "Ant-v5": ("Ant", 1000, dict(use_contact_force=1, post_constraint=1, exclude_worldbody=1,
                                 legacy_healthy_reward=0)),
    "Walker2d-v4": ("Walker2d", 1000, {}),
    "Walker2d-v5": ("Walker2dV5", 1000, dict(post_constraint=1, legacy_healthy_reward=0)),
    "Hopper-v4": ("Hopper", 1000, {}),
    "Hopper-v5": ("Hopper", 1000, dict(post_constraint=1, legacy_healthy_reward=0)),
    "Swimmer-v4": ("Swimmer", 1000, {}),
    "Swimmer-v5": ("Swimmer", 1000, dict(post_constraint=1)),
    "Reacher-v4": ("Reacher", 50, {}),
    "Reacher-v5": ("Reacher", 50, dict(post_constraint=1, reward_after_step=1, obs_include_z=0)),
    "Pusher-v4": ("Pusher", 100, {}),
    "Pusher-v5": ("PusherV5", 100, dict(post_constraint=1, reward_after_step=1,
                                        weighted_reward_info=1)),
    "InvertedPendulum-v4": ("InvertedPendulum", 1000, {}),
    "InvertedPendulum-v5": ("InvertedPendulum", 1000, dict(post_constraint=1,
                                                           reward_if_not_terminated=1)),
    "InvertedDoublePendulum-v4": ("InvertedDoublePendulum", 1000, {}),
    "InvertedDoublePendulum-v5": ("InvertedDoublePendulum", 1000,
                                  dict(post_constraint=1, reward_if_not_terminated=1,
                                       constraint_obs_dim=1)),
    "Humanoid-v3": ("Humanoid", 1000, dict(use_contact_force=1)),
    "Humanoid-v4": ("Humanoid", 1000, {}),
    "Humanoid-v5": ("Humanoid", 1000, dict(use_contact_force=1, post_constraint=1,
                                           exclude_worldbody=1, exclude_root_actuator=1,
                                           legacy_healthy_reward=0)),
    "HumanoidStandup-v4": ("HumanoidStandup", 1000, {}),
    "HumanoidStandup-v5": ("HumanoidStandup", 1000, dict(post_constraint=1, exclude_worldbody=1,
                                                         exclude_root_actuator=1)),
}

_NATIVE_NAMES = {
    "use_contact_force": "use_contact_force", "post_constraint": "post_constraint",
    "legacy_healthy_reward": "legacy_healthy_reward", "reward_after_step": "reward_after_step",
    "obs_include_z": "obs_include_z_distance", "weighted_reward_info": "weighted_reward_info",
    "reward_if_not_terminated": "reward_if_not_terminated", "constraint_obs_dim": "constraint_obs_dim",
    "exclude_root_actuator": "exclude_root_actuator_forces",
    "frame_skip": "frame_skip", "ctrl_cost_weight": "ctrl_cost_weight",
    "forward_reward_weight": "forward_reward_weight", "reset_noise_scale": "reset_noise_scale",
    "dist_cost_weight": "dist_cost_weight", "near_cost_weight": "near_cost_weight",
    "frame_stack": "frame_stack",
}


def native_variant(name, **more):
    """(DevicePool family, params under the reference's config key names) of a GYM_VARIANTS id."""
    task, max_steps, over = GYM_VARIANTS[name]
    over = {**over, **more}
    family = task.replace("V5", "")
    params = {"post_constraint": 0}
    if task.endswith("V5"):
        params["xml_v5"] = 1
    for k, v in over.items():
        if k == "exclude_worldbody":
            params["exclude_worldbody_contact_forces" if family == "Ant"
                   else "exclude_worldbody_observations"] = v
        else:
            params[_NATIVE_NAMES[k]] = v
    if family in ("HalfCheetah", "Walker2d", "Hopper", "Ant"):
        params.setdefault("precision", 1)
    return family, max_steps, params
