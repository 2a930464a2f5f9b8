"""Task ids of envpool/classic_control/registration.py:19-86."""
from envpool_amd.registration import register

_P = "envpool_amd.classic_control"


def _reg(task_id, stem, aliases=(), **kw):
    register(task_id=task_id, import_path=_P, spec_cls=f"{stem}EnvSpec",
             dm_cls=f"{stem}DMEnvPool", gymnasium_cls=f"{stem}GymnasiumEnvPool",
             aliases=list(aliases), **kw)


_reg("CartPole-v0", "CartPole", ["phys2d/CartPole-v0"], max_episode_steps=200,
     reward_threshold=195.0)
_reg("CartPole-v1", "CartPole", ["phys2d/CartPole-v1"], max_episode_steps=500,
     reward_threshold=475.0)
_reg("Pendulum-v0", "Pendulum", ["phys2d/Pendulum-v0"], version=0, max_episode_steps=200)
_reg("Pendulum-v1", "Pendulum", version=1, max_episode_steps=200)
_reg("MountainCar-v0", "MountainCar", max_episode_steps=200)
_reg("MountainCarContinuous-v0", "MountainCarContinuous", max_episode_steps=999)
_reg("Acrobot-v1", "Acrobot", max_episode_steps=500)
