"""Task ids of envpool/mujoco/gym/registration.py:21-93 (hot-path tasks only)."""
from envpool_amd.registration import register

gym_mujoco_envs = [
    ("Ant", ("v3", "v4", "v5"), 1000),
    ("HalfCheetah", ("v3", "v4", "v5"), 1000),
    ("Hopper", ("v3", "v4", "v5"), 1000),
    ("InvertedDoublePendulum", ("v2", "v4", "v5"), 1000),
    ("InvertedPendulum", ("v2", "v4", "v5"), 1000),
    ("Reacher", ("v2", "v4", "v5"), 50),
    ("Swimmer", ("v3", "v4", "v5"), 1000),
    ("Walker2d", ("v3", "v4", "v5"), 1000),
]

for task, versions, max_episode_steps in gym_mujoco_envs:
    for version in versions:
        extra_args = {}
        if version == "v5":
            extra_args["gymnasium_v5_render_camera"] = True
        if task == "Ant" and version == "v3":  # gym/registration.py:39-40
            extra_args["use_contact_force"] = True
        if task == "Ant" and version == "v5":  # gym/registration.py:41-46
            extra_args.update({
                "use_contact_force": True,
                "legacy_healthy_reward": False,
                "exclude_worldbody_contact_forces": True,
            })
        if task == "Hopper" and version == "v5":  # gym/registration.py:47-48
            extra_args["legacy_healthy_reward"] = False
        if task == "InvertedDoublePendulum" and version == "v5":  # gym/registration.py:61-65
            extra_args.update({
                "constraint_obs_dim": 1,
                "reward_if_not_terminated": True,
            })
        if task == "InvertedPendulum" and version == "v5":  # gym/registration.py:66-67
            extra_args["reward_if_not_terminated"] = True
        if task == "Reacher" and version == "v5":  # gym/registration.py:74-78
            extra_args.update({
                "reward_after_step": True,
                "obs_include_z_distance": False,
            })
        if task == "Walker2d" and version == "v5":  # gym/registration.py:79-83
            extra_args.update({
                "xml_file": "walker2d_v5.xml",
                "legacy_healthy_reward": False,
            })
        register(
            task_id=f"{task}-{version}",
            import_path="envpool_amd.mujoco.gym",
            spec_cls=f"Gym{task}EnvSpec",
            dm_cls=f"Gym{task}DMEnvPool",
            gymnasium_cls=f"Gym{task}GymnasiumEnvPool",
            post_constraint=(version == "v5"),
            max_episode_steps=max_episode_steps,
            **extra_args,
        )
