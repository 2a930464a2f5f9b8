"""gym-MuJoCo envs (mirror of envpool/mujoco/gym/__init__.py).

Spec tables restate `HalfCheetahEnvFns` (half_cheetah.h:31-62); the pixel
variants are out of scope.  `precision` is an extension key: 64 (default, the
reference's mjtNum=double) or 32 (fp32 arithmetic, fp64 state and I/O).
"""

import numpy as np

from envpool_amd.core.binding import FamilyDef, make_native_classes, spec
from envpool_amd.python.api import py_env

_inf = float("inf")


def _precision(c):
    if c["precision"] not in (32, 64):
        raise ValueError("precision must be 32 or 64")
    return 1 if c["precision"] == 64 else 0


_HalfCheetah = FamilyDef(
    name="GymHalfCheetah", native="HalfCheetah",
    default_config=[
        ("reward_threshold", 4800.0), ("frame_skip", 5), ("frame_stack", 1),
        ("post_constraint", True),
        ("exclude_current_positions_from_observation", True),
        ("xml_file", "half_cheetah.xml"), ("gymnasium_v5_render_camera", False),
        ("ctrl_cost_weight", 0.1), ("forward_reward_weight", 1.0),
        ("reset_noise_scale", 0.1), ("precision", 64),
    ],
    state_spec=lambda c: [
        ("obs", spec(np.float64,
                     [17 if c["exclude_current_positions_from_observation"] else 18],
                     (-_inf, _inf))),
        ("info:reward_run", spec(np.float64, [-1])),
        ("info:reward_ctrl", spec(np.float64, [-1])),
        ("info:x_position", spec(np.float64, [-1])),
        ("info:x_velocity", spec(np.float64, [-1])),
    ],
    action_spec=lambda c: [("action", spec(np.float64, [-1, 6], (-1.0, 1.0)))],
    native_params=lambda c: {
        "frame_skip": c["frame_skip"],
        "exclude_current_positions_from_observation":
            c["exclude_current_positions_from_observation"],
        "ctrl_cost_weight": c["ctrl_cost_weight"],
        "forward_reward_weight": c["forward_reward_weight"],
        "reset_noise_scale": c["reset_noise_scale"],
        "precision": _precision(c),
    },
    # the model constants are compiled in from half_cheetah_envpool.xml
    unsupported={"frame_stack": 1, "xml_file": "half_cheetah.xml"},
)

_GymHalfCheetahEnvSpec, _GymHalfCheetahEnvPool = make_native_classes(_HalfCheetah)
(GymHalfCheetahEnvSpec, GymHalfCheetahDMEnvPool,
 GymHalfCheetahGymnasiumEnvPool) = py_env(_GymHalfCheetahEnvSpec, _GymHalfCheetahEnvPool)

__all__ = ["GymHalfCheetahEnvSpec", "GymHalfCheetahDMEnvPool",
           "GymHalfCheetahGymnasiumEnvPool"]
