"""Helpers for the `-m gpu` parity tests (call the product through the C ABI)."""
import numpy as np

from envpool_amd.core.device_pool import DevicePool
from oracle_cases import CASES

PARAM_NAMES = {
    "Pendulum": ("version",),
    "FrozenLake": ("size",),
    "CliffWalking": ("is_slippery",),
    "Blackjack": ("natural", "sab"),
    "Catch": ("height", "width"),
}


def make_hip_pool(name, n, seed, extra_params=None, **kw):
    c = CASES[name]
    params = dict(zip(PARAM_NAMES.get(c["task"], ()), c["extra"]))
    params.update(extra_params or {})
    return DevicePool(c["task"], n, seed=seed, max_episode_steps=c["max_steps"],
                      params=params, **kw)


class HipAsOracle:
    """DevicePool with the tiny reset()/step() -> dict surface of oracle.orc."""

    def __init__(self, pool):
        self.pool = pool
        self.ids = np.arange(pool.num_envs, dtype=np.int32)

    def reset(self, ids=None):
        self.pool.reset(self.ids if ids is None else ids)
        return self._flat(self.pool.recv_dict())

    def step(self, action, ids=None):
        self.pool.send(self.ids if ids is None else ids, action)
        return self._flat(self.pool.recv_dict())

    @staticmethod
    def _flat(d):
        return {k: v.reshape(v.shape[0], -1) for k, v in d.items()}
