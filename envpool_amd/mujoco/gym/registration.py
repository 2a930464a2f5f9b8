"""Task ids of the gym-MuJoCo families that have a HIP kernel.

The ids, `max_episode_steps` and the per-version overrides restate
envpool/mujoco/gym/registration.py:21-93; they are kept here as one table so the
version differences can be read at a glance.  `post_constraint` is True for v5 only
(mj_rnePostConstraint after the frame_skip loop, mujoco_env.h:145-147).
"""
from envpool_amd.registration import register

_V5_CAMERA = {"gymnasium_v5_render_camera": True}

# task -> (max_episode_steps, {version: extra config})
_TASKS = {
    "Ant": (1000, {
        "v3": {"use_contact_force": True},
        "v4": {},
        "v5": {"use_contact_force": True, "legacy_healthy_reward": False,
               "exclude_worldbody_contact_forces": True},
    }),
    "HalfCheetah": (1000, {"v3": {}, "v4": {}, "v5": {}}),
    "Hopper": (1000, {"v3": {}, "v4": {}, "v5": {"legacy_healthy_reward": False}}),
    "Humanoid": (1000, {
        "v3": {"use_contact_force": True},
        "v4": {},
        "v5": {"use_contact_force": True, "legacy_healthy_reward": False,
               "exclude_worldbody_observations": True, "exclude_root_actuator_forces": True},
    }),
    "HumanoidStandup": (1000, {
        "v2": {}, "v4": {},
        "v5": {"exclude_worldbody_observations": True, "exclude_root_actuator_forces": True},
    }),
    "InvertedDoublePendulum": (1000, {
        "v2": {}, "v4": {},
        "v5": {"constraint_obs_dim": 1, "reward_if_not_terminated": True},
    }),
    "InvertedPendulum": (1000, {"v2": {}, "v4": {}, "v5": {"reward_if_not_terminated": True}}),
    "Pusher": (100, {
        "v2": {}, "v4": {},
        "v5": {"xml_file": "pusher_v5.xml", "reward_after_step": True,
               "weighted_reward_info": True},
    }),
    "Reacher": (50, {
        "v2": {}, "v4": {},
        "v5": {"reward_after_step": True, "obs_include_z_distance": False},
    }),
    "Swimmer": (1000, {"v3": {}, "v4": {}, "v5": {}}),
    "Walker2d": (1000, {
        "v3": {}, "v4": {},
        "v5": {"xml_file": "walker2d_v5.xml", "legacy_healthy_reward": False},
    }),
}

for _task, (_steps, _versions) in _TASKS.items():
    for _version, _extra in _versions.items():
        _cfg = dict(_V5_CAMERA) if _version == "v5" else {}
        _cfg.update(_extra)
        register(
            task_id=f"{_task}-{_version}",
            import_path="envpool_amd.mujoco.gym",
            spec_cls=f"Gym{_task}EnvSpec",
            dm_cls=f"Gym{_task}DMEnvPool",
            gymnasium_cls=f"Gym{_task}GymnasiumEnvPool",
            post_constraint=_version == "v5",
            max_episode_steps=_steps,
            **_cfg,
        )
