"""gym-MuJoCo envs (mirror of envpool/mujoco/gym/__init__.py).

Spec tables restate `HalfCheetahEnvFns` (half_cheetah.h:31-62), `AntEnvFns`
(ant.h:31-75; v3/v5 add 6 contact-force numbers per body) and `Walker2dEnvFns`
(walker2d.h:30-67), `InvertedPendulumEnvFns` (inverted_pendulum.h:30-60) and
`InvertedDoublePendulumEnvFns` (inverted_double_pendulum.h:30-62) and
`ReacherEnvFns` (reacher.h:30-65), `SwimmerEnvFns` (swimmer.h:30-66), `HopperEnvFns` (hopper.h:30-70),
`HumanoidEnvFns` (humanoid.h:30-82), `HumanoidStandupEnvFns` (humanoid_standup.h:30-73); the pixel
variants are out of scope.  `precision` is an extension key: 64 (default, the
reference's mjtNum=double); Ant also accepts 32 (fp32 arithmetic, fp64 state and I/O, within
1e-5).  The planar families' fp32 mode was removed in round 4 (outside 1e-5, slower than fp64).
"""

import numpy as np

from envpool_amd.core.binding import FamilyDef, make_native_classes, spec
from envpool_amd.python.api import py_env

_inf = float("inf")


def _stack(shape, c):
    """StackSpec (envpool/mujoco/frame_stack.h:42-71)."""
    if c["frame_stack"] < 1:
        raise ValueError("frame_stack must be greater than 0")
    return [c["frame_stack"], *shape] if c["frame_stack"] > 1 else shape


def _precision(c):
    if c["precision"] not in (32, 64):
        raise ValueError("precision must be 32 or 64")
    return 1 if c["precision"] == 64 else 0


def _precision64(c):
    """HalfCheetah / Walker2d / Hopper: fp64 only (their fp32 arithmetic mode is gone)."""
    if c["precision"] != 64:
        raise ValueError("precision must be 64 for this family (the fp32 mode of the planar "
                         "kernels was removed: outside 1e-5 and slower than fp64)")
    return 1


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
                     _stack([17 if c["exclude_current_positions_from_observation"] else 18], c),
                     (-_inf, _inf))),
        ("info:reward_run", spec(np.float64, [-1])),
        ("info:reward_ctrl", spec(np.float64, [-1])),
        ("info:x_position", spec(np.float64, [-1])),
        ("info:x_velocity", spec(np.float64, [-1])),
    ],
    action_spec=lambda c: [("action", spec(np.float64, [-1, 6], (-1.0, 1.0)))],
    native_params=lambda c: {
        "frame_skip": c["frame_skip"],
        "frame_stack": c["frame_stack"],
        "exclude_current_positions_from_observation":
            c["exclude_current_positions_from_observation"],
        "ctrl_cost_weight": c["ctrl_cost_weight"],
        "forward_reward_weight": c["forward_reward_weight"],
        "reset_noise_scale": c["reset_noise_scale"],
        "precision": _precision64(c),
    },
    # the model constants are compiled in from half_cheetah_envpool.xml
    unsupported={"xml_file": "half_cheetah.xml"},
)

_Ant = FamilyDef(
    name="GymAnt", native="Ant",
    # ant.h:33-50
    default_config=[
        ("reward_threshold", 6000.0), ("frame_skip", 5), ("frame_stack", 1),
        ("post_constraint", True), ("use_contact_force", False),
        ("legacy_healthy_reward", True), ("exclude_worldbody_contact_forces", False),
        ("terminate_when_unhealthy", True),
        ("exclude_current_positions_from_observation", True),
        ("xml_file", "ant.xml"), ("gymnasium_v5_render_camera", False),
        ("forward_reward_weight", 1.0), ("ctrl_cost_weight", 0.5),
        ("contact_cost_weight", 5e-4), ("healthy_reward", 1.0),
        ("healthy_z_min", 0.2), ("healthy_z_max", 1.0),
        ("contact_force_min", -1.0), ("contact_force_max", 1.0),
        ("reset_noise_scale", 0.1), ("precision", 64),
    ],
    state_spec=lambda c: [
        # ant.h:51-66: + 6 per MuJoCo body (14, world optional) with use_contact_force
        ("obs", spec(np.float64,
                     _stack([(27 if c["exclude_current_positions_from_observation"] else 29) +
                             (6 * (14 - (1 if c["exclude_worldbody_contact_forces"] else 0))
                              if c["use_contact_force"] else 0)], c),
                     (-_inf, _inf))),
    ] + [(k, spec(np.float64, [-1])) for k in (
        "info:reward_forward", "info:reward_ctrl", "info:reward_contact",
        "info:reward_survive", "info:x_position", "info:y_position",
        "info:distance_from_origin", "info:x_velocity", "info:y_velocity")],
    action_spec=lambda c: [("action", spec(np.float64, [-1, 8], (-1.0, 1.0)))],
    native_params=lambda c: {
        "frame_skip": c["frame_skip"],
        "frame_stack": c["frame_stack"],
        "exclude_current_positions_from_observation":
            c["exclude_current_positions_from_observation"],
        "terminate_when_unhealthy": c["terminate_when_unhealthy"],
        "legacy_healthy_reward": c["legacy_healthy_reward"],
        "ctrl_cost_weight": c["ctrl_cost_weight"],
        "forward_reward_weight": c["forward_reward_weight"],
        "healthy_reward": c["healthy_reward"],
        "healthy_z_min": c["healthy_z_min"], "healthy_z_max": c["healthy_z_max"],
        "reset_noise_scale": c["reset_noise_scale"],
        "use_contact_force": c["use_contact_force"],
        "post_constraint": c["post_constraint"],
        "exclude_worldbody_contact_forces": c["exclude_worldbody_contact_forces"],
        "contact_cost_weight": c["contact_cost_weight"],
        "contact_force_min": c["contact_force_min"],
        "contact_force_max": c["contact_force_max"],
        "precision": _precision(c),
    },
    unsupported={"xml_file": "ant.xml"},
)



def _walker_xml(c):
    # the model constants are compiled in from walker2d_envpool.xml /
    # walker2d_v5_envpool.xml (mujoco_env.h:50-58 resolves them from these names)
    if c["xml_file"] not in ("walker2d.xml", "walker2d_v5.xml"):
        raise ValueError(
            f"GymWalker2d: xml_file={c['xml_file']!r} is not supported by the MI355X "
            "engine (only 'walker2d.xml' and 'walker2d_v5.xml')")
    return 1 if c["xml_file"] == "walker2d_v5.xml" else 0


_Walker2d = FamilyDef(
    name="GymWalker2d", native="Walker2d",
    # walker2d.h:32-47
    default_config=[
        ("frame_skip", 4), ("frame_stack", 1), ("post_constraint", True),
        ("ctrl_cost_weight", 0.001), ("terminate_when_unhealthy", True),
        ("exclude_current_positions_from_observation", True),
        ("legacy_healthy_reward", True), ("xml_file", "walker2d.xml"),
        ("gymnasium_v5_render_camera", False),
        ("forward_reward_weight", 1.0), ("healthy_reward", 1.0),
        ("healthy_z_min", 0.8), ("healthy_z_max", 2.0),
        ("healthy_angle_min", -1.0), ("healthy_angle_max", 1.0),
        ("velocity_min", -10.0), ("velocity_max", 10.0),
        ("reset_noise_scale", 0.005), ("precision", 64),
    ],
    state_spec=lambda c: [
        ("obs", spec(np.float64,
                     _stack([17 if c["exclude_current_positions_from_observation"] else 18], c),
                     (-_inf, _inf))),
        ("info:x_position", spec(np.float64, [-1])),
        ("info:x_velocity", spec(np.float64, [-1])),
    ],
    action_spec=lambda c: [("action", spec(np.float64, [-1, 6], (-1.0, 1.0)))],
    native_params=lambda c: {
        "frame_skip": c["frame_skip"],
        "frame_stack": c["frame_stack"],
        "exclude_current_positions_from_observation":
            c["exclude_current_positions_from_observation"],
        "terminate_when_unhealthy": c["terminate_when_unhealthy"],
        "legacy_healthy_reward": c["legacy_healthy_reward"],
        "ctrl_cost_weight": c["ctrl_cost_weight"],
        "forward_reward_weight": c["forward_reward_weight"],
        "healthy_reward": c["healthy_reward"],
        "healthy_z_min": c["healthy_z_min"], "healthy_z_max": c["healthy_z_max"],
        "healthy_angle_min": c["healthy_angle_min"],
        "healthy_angle_max": c["healthy_angle_max"],
        "velocity_min": c["velocity_min"], "velocity_max": c["velocity_max"],
        "reset_noise_scale": c["reset_noise_scale"],
        "xml_v5": _walker_xml(c),
        "precision": _precision64(c),
    },
)

_InvertedPendulum = FamilyDef(
    name="GymInvertedPendulum", native="InvertedPendulum",
    # inverted_pendulum.h:32-41
    default_config=[
        ("reward_threshold", 950.0), ("frame_skip", 2), ("frame_stack", 1),
        ("post_constraint", True), ("healthy_reward", 1.0),
        ("reward_if_not_terminated", False), ("xml_file", "inverted_pendulum.xml"),
        ("gymnasium_v5_render_camera", False), ("healthy_z_min", -0.2),
        ("healthy_z_max", 0.2), ("reset_noise_scale", 0.01),
    ],
    state_spec=lambda c: [("obs", spec(np.float64, _stack([4], c), (-_inf, _inf)))],
    action_spec=lambda c: [("action", spec(np.float64, [-1, 1], (-3.0, 3.0)))],
    native_params=lambda c: {
        "frame_skip": c["frame_skip"], "frame_stack": c["frame_stack"],
        "healthy_reward": c["healthy_reward"],
        "reward_if_not_terminated": c["reward_if_not_terminated"],
        "healthy_z_min": c["healthy_z_min"], "healthy_z_max": c["healthy_z_max"],
        "reset_noise_scale": c["reset_noise_scale"],
    },
    unsupported={"xml_file": "inverted_pendulum.xml"},
)


def _constraint_obs_dim(c):
    if not 0 <= c["constraint_obs_dim"] <= 3:
        raise ValueError("constraint_obs_dim must be in [0, 3]")
    return c["constraint_obs_dim"]


_InvertedDoublePendulum = FamilyDef(
    name="GymInvertedDoublePendulum", native="InvertedDoublePendulum",
    # inverted_double_pendulum.h:32-44
    default_config=[
        ("reward_threshold", 9100.0), ("frame_skip", 5), ("frame_stack", 1),
        ("post_constraint", True), ("healthy_reward", 10.0),
        ("reward_if_not_terminated", False), ("constraint_obs_dim", 3),
        ("xml_file", "inverted_double_pendulum.xml"),
        ("gymnasium_v5_render_camera", False), ("healthy_z_max", 1.0),
        ("observation_min", -10.0), ("observation_max", 10.0),
        ("reset_noise_scale", 0.1),
    ],
    state_spec=lambda c: [
        ("obs", spec(np.float64, _stack([8 + _constraint_obs_dim(c)], c), (-_inf, _inf)))],
    action_spec=lambda c: [("action", spec(np.float64, [-1, 1], (-1.0, 1.0)))],
    native_params=lambda c: {
        "frame_skip": c["frame_skip"], "frame_stack": c["frame_stack"],
        "healthy_reward": c["healthy_reward"],
        "reward_if_not_terminated": c["reward_if_not_terminated"],
        "constraint_obs_dim": _constraint_obs_dim(c),
        "healthy_z_max": c["healthy_z_max"],
        "observation_min": c["observation_min"], "observation_max": c["observation_max"],
        "reset_noise_scale": c["reset_noise_scale"],
    },
    unsupported={"xml_file": "inverted_double_pendulum.xml"},
)

_Reacher = FamilyDef(
    name="GymReacher", native="Reacher",
    # reacher.h:32-43
    default_config=[
        ("reward_threshold", -3.75), ("frame_skip", 2), ("frame_stack", 1),
        ("post_constraint", True), ("ctrl_cost_weight", 1.0),
        ("reward_after_step", False), ("obs_include_z_distance", True),
        ("dist_cost_weight", 1.0), ("xml_file", "reacher.xml"),
        ("gymnasium_v5_render_camera", False), ("reset_qpos_scale", 0.1),
        ("reset_qvel_scale", 0.005), ("reset_goal_scale", 0.2),
    ],
    state_spec=lambda c: [
        ("obs", spec(np.float64, _stack([11 if c["obs_include_z_distance"] else 10], c),
                     (-_inf, _inf))),
        ("info:reward_dist", spec(np.float64, [-1])),
        ("info:reward_ctrl", spec(np.float64, [-1])),
    ],
    action_spec=lambda c: [("action", spec(np.float64, [-1, 2], (-1.0, 1.0)))],
    native_params=lambda c: {
        "frame_skip": c["frame_skip"], "frame_stack": c["frame_stack"],
        "ctrl_cost_weight": c["ctrl_cost_weight"],
        "reward_after_step": c["reward_after_step"],
        "obs_include_z_distance": c["obs_include_z_distance"],
        "dist_cost_weight": c["dist_cost_weight"],
        "reset_qpos_scale": c["reset_qpos_scale"],
        "reset_qvel_scale": c["reset_qvel_scale"],
        "reset_goal_scale": c["reset_goal_scale"],
    },
    unsupported={"xml_file": "reacher.xml"},
)

def _pusher_xml(c: dict) -> int:
    # pusher_envpool.xml / pusher_v5_envpool.xml (mujoco_env.h:50-58 resolves them from these names)
    if c["xml_file"] not in ("pusher.xml", "pusher_v5.xml"):
        raise ValueError(
            f"GymPusher: xml_file={c['xml_file']!r} is not supported by the MI355X "
            "engine (only 'pusher.xml' and 'pusher_v5.xml')")
    return 1 if c["xml_file"] == "pusher_v5.xml" else 0


_Pusher = FamilyDef(
    name="GymPusher", native="Pusher",
    # pusher.h:33-46
    default_config=[
        ("reward_threshold", 0.0), ("frame_skip", 5), ("frame_stack", 1),
        ("post_constraint", True), ("ctrl_cost_weight", 0.1), ("dist_cost_weight", 1.0),
        ("near_cost_weight", 0.5), ("xml_file", "pusher.xml"),
        ("gymnasium_v5_render_camera", False), ("reward_after_step", False),
        ("weighted_reward_info", False), ("reset_qvel_scale", 0.005),
        ("cylinder_x_min", -0.3), ("cylinder_x_max", 0.0), ("cylinder_y_min", -0.2),
        ("cylinder_y_max", 0.2), ("cylinder_dist_min", 0.17),
    ],
    state_spec=lambda c: [
        ("obs", spec(np.float64, _stack([23], c), (-_inf, _inf))),
        ("info:reward_dist", spec(np.float64, [-1])),
        ("info:reward_ctrl", spec(np.float64, [-1])),
        ("info:reward_near", spec(np.float64, [-1])),
    ],
    action_spec=lambda c: [("action", spec(np.float64, [-1, 7], (-2.0, 2.0)))],
    native_params=lambda c: {
        "frame_skip": c["frame_skip"], "frame_stack": c["frame_stack"],
        "ctrl_cost_weight": c["ctrl_cost_weight"], "dist_cost_weight": c["dist_cost_weight"],
        "near_cost_weight": c["near_cost_weight"],
        "reward_after_step": c["reward_after_step"],
        "weighted_reward_info": c["weighted_reward_info"],
        "reset_qvel_scale": c["reset_qvel_scale"],
        "cylinder_x_min": c["cylinder_x_min"], "cylinder_x_max": c["cylinder_x_max"],
        "cylinder_y_min": c["cylinder_y_min"], "cylinder_y_max": c["cylinder_y_max"],
        "cylinder_dist_min": c["cylinder_dist_min"],
        "xml_v5": _pusher_xml(c),
    },
)

_Swimmer = FamilyDef(
    name="GymSwimmer", native="Swimmer",
    # swimmer.h:32-42
    default_config=[
        ("reward_threshold", 360.0), ("frame_skip", 4), ("frame_stack", 1),
        ("post_constraint", True),
        ("exclude_current_positions_from_observation", True),
        ("xml_file", "swimmer.xml"), ("gymnasium_v5_render_camera", False),
        ("forward_reward_weight", 1.0), ("ctrl_cost_weight", 1e-4),
        ("reset_noise_scale", 0.1),
    ],
    state_spec=lambda c: [
        ("obs", spec(np.float64,
                     _stack([8 if c["exclude_current_positions_from_observation"] else 10], c),
                     (-_inf, _inf))),
    ] + [(k, spec(np.float64, [-1])) for k in (
        "info:reward_fwd", "info:reward_ctrl", "info:x_position", "info:y_position",
        "info:distance_from_origin", "info:x_velocity", "info:y_velocity")],
    action_spec=lambda c: [("action", spec(np.float64, [-1, 2], (-1.0, 1.0)))],
    native_params=lambda c: {
        "frame_skip": c["frame_skip"], "frame_stack": c["frame_stack"],
        "exclude_current_positions_from_observation":
            c["exclude_current_positions_from_observation"],
        "forward_reward_weight": c["forward_reward_weight"],
        "ctrl_cost_weight": c["ctrl_cost_weight"],
        "reset_noise_scale": c["reset_noise_scale"],
    },
    unsupported={"xml_file": "swimmer.xml"},
)

_Hopper = FamilyDef(
    name="GymHopper", native="Hopper",
    # hopper.h:32-49
    default_config=[
        ("reward_threshold", 6000.0), ("frame_skip", 4), ("frame_stack", 1),
        ("post_constraint", True), ("terminate_when_unhealthy", True),
        ("legacy_healthy_reward", True),
        ("exclude_current_positions_from_observation", True),
        ("xml_file", "hopper.xml"), ("gymnasium_v5_render_camera", False),
        ("ctrl_cost_weight", 1e-3), ("forward_reward_weight", 1.0),
        ("healthy_reward", 1.0), ("velocity_min", -10.0), ("velocity_max", 10.0),
        ("healthy_state_min", -100.0), ("healthy_state_max", 100.0),
        ("healthy_angle_min", -0.2), ("healthy_angle_max", 0.2),
        ("healthy_z_min", 0.7), ("reset_noise_scale", 5e-3), ("precision", 64),
    ],
    state_spec=lambda c: [
        ("obs", spec(np.float64,
                     _stack([11 if c["exclude_current_positions_from_observation"] else 12], c),
                     (-_inf, _inf))),
        ("info:x_position", spec(np.float64, [-1])),
        ("info:x_velocity", spec(np.float64, [-1])),
    ],
    action_spec=lambda c: [("action", spec(np.float64, [-1, 3], (-1.0, 1.0)))],
    native_params=lambda c: {
        "frame_skip": c["frame_skip"], "frame_stack": c["frame_stack"],
        "exclude_current_positions_from_observation":
            c["exclude_current_positions_from_observation"],
        "terminate_when_unhealthy": c["terminate_when_unhealthy"],
        "legacy_healthy_reward": c["legacy_healthy_reward"],
        "ctrl_cost_weight": c["ctrl_cost_weight"],
        "forward_reward_weight": c["forward_reward_weight"],
        "healthy_reward": c["healthy_reward"],
        "velocity_min": c["velocity_min"], "velocity_max": c["velocity_max"],
        "healthy_state_min": c["healthy_state_min"],
        "healthy_state_max": c["healthy_state_max"],
        "healthy_angle_min": c["healthy_angle_min"],
        "healthy_angle_max": c["healthy_angle_max"],
        "healthy_z_min": c["healthy_z_min"],
        "reset_noise_scale": c["reset_noise_scale"],
        "precision": _precision64(c),
    },
    unsupported={"xml_file": "hopper.xml"},
)

def _humanoid_obs_dim(c):
    """humanoid.h:50-60."""
    n = 376 if c["exclude_current_positions_from_observation"] else 378
    if c["exclude_worldbody_observations"]:
        n -= 10 + 6 + 6
    if c["exclude_root_actuator_forces"]:
        n -= 6
    return n


def _humanoid_params(c, keys):
    return {k: c[k] for k in keys}


_HUMANOID_SHARED = ("frame_skip", "frame_stack", "post_constraint",
                    "exclude_current_positions_from_observation",
                    "exclude_worldbody_observations", "exclude_root_actuator_forces",
                    "forward_reward_weight", "ctrl_cost_weight", "healthy_reward",
                    "contact_cost_weight", "contact_cost_max", "reset_noise_scale")

_Humanoid = FamilyDef(
    name="GymHumanoid", native="Humanoid",
    # humanoid.h:32-48
    default_config=[
        ("frame_skip", 5), ("frame_stack", 1), ("post_constraint", True),
        ("legacy_healthy_reward", True), ("exclude_worldbody_observations", False),
        ("exclude_root_actuator_forces", False), ("use_contact_force", False),
        ("forward_reward_weight", 1.25), ("terminate_when_unhealthy", True),
        ("exclude_current_positions_from_observation", True),
        ("xml_file", "humanoid.xml"), ("gymnasium_v5_render_camera", False),
        ("ctrl_cost_weight", 0.1), ("healthy_reward", 5.0), ("healthy_z_min", 1.0),
        ("healthy_z_max", 2.0), ("contact_cost_weight", 5e-7), ("contact_cost_max", 10.0),
        ("reset_noise_scale", 1e-2),
    ],
    state_spec=lambda c: [
        ("obs", spec(np.float64, _stack([_humanoid_obs_dim(c)], c), (-_inf, _inf))),
        ("info:reward_linvel", spec(np.float64, [-1])),
        ("info:reward_quadctrl", spec(np.float64, [-1])),
        ("info:reward_alive", spec(np.float64, [-1])),
        ("info:reward_impact", spec(np.float64, [-1])),
        ("info:x_position", spec(np.float64, [-1])),
        ("info:y_position", spec(np.float64, [-1])),
        ("info:distance_from_origin", spec(np.float64, [-1])),
        ("info:x_velocity", spec(np.float64, [-1])),
        ("info:y_velocity", spec(np.float64, [-1])),
    ],
    action_spec=lambda c: [("action", spec(np.float64, [-1, 17], (-0.4, 0.4)))],
    native_params=lambda c: _humanoid_params(
        c, _HUMANOID_SHARED + ("legacy_healthy_reward", "use_contact_force",
                               "terminate_when_unhealthy", "healthy_z_min", "healthy_z_max")),
    unsupported={"xml_file": "humanoid.xml"},
)

_HumanoidStandup = FamilyDef(
    name="GymHumanoidStandup", native="HumanoidStandup",
    # humanoid_standup.h:32-45
    default_config=[
        ("frame_skip", 5), ("frame_stack", 1), ("post_constraint", True),
        ("forward_reward_weight", 1.0),
        ("exclude_current_positions_from_observation", True),
        ("exclude_worldbody_observations", False), ("exclude_root_actuator_forces", False),
        ("xml_file", "humanoidstandup.xml"), ("gymnasium_v5_render_camera", False),
        ("ctrl_cost_weight", 0.1), ("contact_cost_weight", 5e-7), ("contact_cost_max", 10.0),
        ("healthy_reward", 1.0), ("reset_noise_scale", 1e-2),
    ],
    # humanoid_standup.h:47-66: `obs` first, then the four reward terms
    state_spec=lambda c: [
        ("obs", spec(np.float64, _stack([_humanoid_obs_dim(c)], c), (-_inf, _inf))),
        ("info:reward_linup", spec(np.float64, [-1])),
        ("info:reward_quadctrl", spec(np.float64, [-1])),
        ("info:reward_alive", spec(np.float64, [-1])),
        ("info:reward_impact", spec(np.float64, [-1])),
    ],
    action_spec=lambda c: [("action", spec(np.float64, [-1, 17], (-0.4, 0.4)))],
    native_params=lambda c: _humanoid_params(c, _HUMANOID_SHARED),
    unsupported={"xml_file": "humanoidstandup.xml"},
)

_GymHalfCheetahEnvSpec, _GymHalfCheetahEnvPool = make_native_classes(_HalfCheetah)
_GymHumanoidEnvSpec, _GymHumanoidEnvPool = make_native_classes(_Humanoid)
(GymHumanoidEnvSpec, GymHumanoidDMEnvPool,
 GymHumanoidGymnasiumEnvPool) = py_env(_GymHumanoidEnvSpec, _GymHumanoidEnvPool)
_GymHumanoidStandupEnvSpec, _GymHumanoidStandupEnvPool = make_native_classes(_HumanoidStandup)
(GymHumanoidStandupEnvSpec, GymHumanoidStandupDMEnvPool,
 GymHumanoidStandupGymnasiumEnvPool) = py_env(_GymHumanoidStandupEnvSpec,
                                              _GymHumanoidStandupEnvPool)
_GymHopperEnvSpec, _GymHopperEnvPool = make_native_classes(_Hopper)
(GymHopperEnvSpec, GymHopperDMEnvPool,
 GymHopperGymnasiumEnvPool) = py_env(_GymHopperEnvSpec, _GymHopperEnvPool)
_GymSwimmerEnvSpec, _GymSwimmerEnvPool = make_native_classes(_Swimmer)
(GymSwimmerEnvSpec, GymSwimmerDMEnvPool,
 GymSwimmerGymnasiumEnvPool) = py_env(_GymSwimmerEnvSpec, _GymSwimmerEnvPool)
_GymPusherEnvSpec, _GymPusherEnvPool = make_native_classes(_Pusher)
(GymPusherEnvSpec, GymPusherDMEnvPool,
 GymPusherGymnasiumEnvPool) = py_env(_GymPusherEnvSpec, _GymPusherEnvPool)
_GymReacherEnvSpec, _GymReacherEnvPool = make_native_classes(_Reacher)
(GymReacherEnvSpec, GymReacherDMEnvPool,
 GymReacherGymnasiumEnvPool) = py_env(_GymReacherEnvSpec, _GymReacherEnvPool)
_GymInvertedPendulumEnvSpec, _GymInvertedPendulumEnvPool = make_native_classes(_InvertedPendulum)
(GymInvertedPendulumEnvSpec, GymInvertedPendulumDMEnvPool,
 GymInvertedPendulumGymnasiumEnvPool) = py_env(_GymInvertedPendulumEnvSpec,
                                               _GymInvertedPendulumEnvPool)
(_GymInvertedDoublePendulumEnvSpec,
 _GymInvertedDoublePendulumEnvPool) = make_native_classes(_InvertedDoublePendulum)
(GymInvertedDoublePendulumEnvSpec, GymInvertedDoublePendulumDMEnvPool,
 GymInvertedDoublePendulumGymnasiumEnvPool) = py_env(_GymInvertedDoublePendulumEnvSpec,
                                                     _GymInvertedDoublePendulumEnvPool)
_GymWalker2dEnvSpec, _GymWalker2dEnvPool = make_native_classes(_Walker2d)
(GymWalker2dEnvSpec, GymWalker2dDMEnvPool,
 GymWalker2dGymnasiumEnvPool) = py_env(_GymWalker2dEnvSpec, _GymWalker2dEnvPool)
_GymAntEnvSpec, _GymAntEnvPool = make_native_classes(_Ant)
GymAntEnvSpec, GymAntDMEnvPool, GymAntGymnasiumEnvPool = py_env(_GymAntEnvSpec, _GymAntEnvPool)
(GymHalfCheetahEnvSpec, GymHalfCheetahDMEnvPool,
 GymHalfCheetahGymnasiumEnvPool) = py_env(_GymHalfCheetahEnvSpec, _GymHalfCheetahEnvPool)

__all__ = ["GymHalfCheetahEnvSpec", "GymHalfCheetahDMEnvPool",
           "GymHalfCheetahGymnasiumEnvPool", "GymAntEnvSpec", "GymAntDMEnvPool",
           "GymAntGymnasiumEnvPool", "GymWalker2dEnvSpec", "GymWalker2dDMEnvPool",
           "GymWalker2dGymnasiumEnvPool", "GymInvertedPendulumEnvSpec",
           "GymInvertedPendulumDMEnvPool", "GymInvertedPendulumGymnasiumEnvPool",
           "GymInvertedDoublePendulumEnvSpec", "GymInvertedDoublePendulumDMEnvPool",
           "GymInvertedDoublePendulumGymnasiumEnvPool", "GymReacherEnvSpec",
           "GymReacherDMEnvPool", "GymReacherGymnasiumEnvPool", "GymPusherEnvSpec",
           "GymPusherDMEnvPool", "GymPusherGymnasiumEnvPool", "GymSwimmerEnvSpec",
           "GymSwimmerDMEnvPool", "GymSwimmerGymnasiumEnvPool", "GymHopperEnvSpec",
           "GymHopperDMEnvPool", "GymHopperGymnasiumEnvPool", "GymHumanoidEnvSpec",
           "GymHumanoidDMEnvPool", "GymHumanoidGymnasiumEnvPool",
           "GymHumanoidStandupEnvSpec", "GymHumanoidStandupDMEnvPool",
           "GymHumanoidStandupGymnasiumEnvPool"]
