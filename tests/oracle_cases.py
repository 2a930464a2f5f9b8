"""Shared table of oracle/parity cases (test infrastructure)."""
import numpy as np

# task family, registered id it mirrors, max_episode_steps, extra cfg, n_actions
# (0 => continuous), action range
CASES = {
    "CartPole-v1": dict(task="CartPole", max_steps=500, extra=(), act=("int", 2)),
    "CartPole-v0": dict(task="CartPole", max_steps=200, extra=(), act=("int", 2)),
    "Pendulum-v0": dict(task="Pendulum", max_steps=200, extra=(0,), act=("float", 2.5)),
    "Pendulum-v1": dict(task="Pendulum", max_steps=200, extra=(1,), act=("float", 2.5)),
    "MountainCar-v0": dict(task="MountainCar", max_steps=200, extra=(), act=("int", 3)),
    "MountainCarContinuous-v0": dict(
        task="MountainCarContinuous", max_steps=999, extra=(), act=("float", 1.3)
    ),
    "Acrobot-v1": dict(task="Acrobot", max_steps=500, extra=(), act=("int", 3)),
    "Catch-v0": dict(task="Catch", max_steps=0, extra=(10, 5), act=("int", 3)),
    "FrozenLake-v1": dict(task="FrozenLake", max_steps=100, extra=(4,), act=("int", 4)),
    "FrozenLake8x8-v1": dict(task="FrozenLake", max_steps=200, extra=(8,), act=("int", 4)),
    "Taxi-v3": dict(task="Taxi", max_steps=200, extra=(), act=("int", 6)),
    "NChain-v0": dict(task="NChain", max_steps=1000, extra=(), act=("int", 2)),
    "CliffWalking-v0": dict(task="CliffWalking", max_steps=0, extra=(0,), act=("int", 4)),
    "CliffWalkingSlippery-v1": dict(
        task="CliffWalking", max_steps=0, extra=(1,), act=("int", 4)
    ),
    "Blackjack-v1": dict(task="Blackjack", max_steps=0, extra=(0, 1), act=("int", 2)),
}

INTEGER_EXACT = {
    "Catch-v0", "FrozenLake-v1", "FrozenLake8x8-v1", "Taxi-v3", "NChain-v0",
    "CliffWalking-v0", "CliffWalkingSlippery-v1", "Blackjack-v1",
}


def sample_actions(case, rng, n):
    kind, p = case["act"]
    if kind == "int":
        return rng.integers(0, p, size=n).astype(np.int32)
    return rng.uniform(-p, p, size=(n, 1)).astype(np.float32)
