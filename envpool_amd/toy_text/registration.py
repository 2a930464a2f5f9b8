"""Task ids of envpool/toy_text/registration.py:17-104."""
from envpool_amd.registration import register

_P = "envpool_amd.toy_text"


def _reg(task_id, stem, aliases=(), **kw):
    register(task_id=task_id, import_path=_P, spec_cls=f"{stem}EnvSpec",
             dm_cls=f"{stem}DMEnvPool", gymnasium_cls=f"{stem}GymnasiumEnvPool",
             aliases=list(aliases), **kw)


_reg("Catch-v0", "Catch", height=10, width=5)
_reg("FrozenLake-v1", "FrozenLake", size=4, max_episode_steps=100, reward_threshold=0.7)
_reg("FrozenLake8x8-v1", "FrozenLake", size=8, max_episode_steps=200, reward_threshold=0.85)
_reg("Taxi-v3", "Taxi", max_episode_steps=200, reward_threshold=8.0)
_reg("NChain-v0", "NChain", max_episode_steps=1000)
_reg("CliffWalking-v1", "CliffWalking", ["tabular/CliffWalking-v0"], is_slippery=False)
_reg("CliffWalkingSlippery-v1", "CliffWalking", is_slippery=True)
_reg("CliffWalking-v0", "CliffWalking", is_slippery=False)
_reg("Blackjack-v1", "Blackjack", ["tabular/Blackjack-v0"], sab=True, natural=False)
