"""Toy text envs (mirror of envpool/toy_text/__init__.py).

Spec tables restate `XxxEnvFns`: catch.h:31-45, frozen_lake.h:32-46,
taxi.h:30-44, nchain.h:31-43, cliffwalking.h:32-46, blackjack.h:31-45.
"""

import numpy as np

from envpool_amd.core.binding import FamilyDef, make_native_classes, spec
from envpool_amd.python.api import py_env

_Catch = FamilyDef(
    name="Catch", native="Catch",
    default_config=[("height", 10), ("width", 5)],
    state_spec=lambda c: [("obs", spec(np.float32, [c["height"], c["width"]], (0.0, 1.0)))],
    action_spec=lambda c: [("action", spec(np.int32, [-1], (0, 2)))],
    native_params=lambda c: {"height": c["height"], "width": c["width"]},
)
_FrozenLake = FamilyDef(
    name="FrozenLake", native="FrozenLake",
    default_config=[("reward_threshold", 0.7), ("size", 4)],
    state_spec=lambda c: [("obs", spec(np.int32, [-1], (0, c["size"] * c["size"] - 1)))],
    action_spec=lambda c: [("action", spec(np.int32, [-1], (0, 3)))],
    native_params=lambda c: {"size": c["size"]},
)
_Taxi = FamilyDef(
    name="Taxi", native="Taxi",
    default_config=[("reward_threshold", 8.0)],
    state_spec=lambda c: [("obs", spec(np.int32, [-1], (0, 499)))],
    action_spec=lambda c: [("action", spec(np.int32, [-1], (0, 5)))],
)
_NChain = FamilyDef(
    name="NChain", native="NChain",
    default_config=[],
    state_spec=lambda c: [("obs", spec(np.int32, [-1], (0, 4)))],
    action_spec=lambda c: [("action", spec(np.int32, [-1], (0, 1)))],
)
_CliffWalking = FamilyDef(
    name="CliffWalking", native="CliffWalking",
    default_config=[("is_slippery", False)],
    state_spec=lambda c: [("obs", spec(np.int32, [-1], (0, 47))),
                          ("info:prob", spec(np.float32, [-1]))],
    action_spec=lambda c: [("action", spec(np.int32, [-1], (0, 3)))],
    native_params=lambda c: {"is_slippery": c["is_slippery"]},
)
_Blackjack = FamilyDef(
    name="Blackjack", native="Blackjack",
    default_config=[("natural", False), ("sab", True)],
    state_spec=lambda c: [("obs", spec(np.int32, [3], (0, 31)))],
    action_spec=lambda c: [("action", spec(np.int32, [-1], (0, 1)))],
    native_params=lambda c: {"natural": c["natural"], "sab": c["sab"]},
)

_g = globals()
__all__ = []
for _fd in (_Catch, _FrozenLake, _Taxi, _NChain, _CliffWalking, _Blackjack):
    _s, _p = make_native_classes(_fd)
    _g[f"_{_fd.name}EnvSpec"], _g[f"_{_fd.name}EnvPool"] = _s, _p
    (_g[f"{_fd.name}EnvSpec"], _g[f"{_fd.name}DMEnvPool"],
     _g[f"{_fd.name}GymnasiumEnvPool"]) = py_env(_s, _p)
    __all__ += [f"{_fd.name}EnvSpec", f"{_fd.name}DMEnvPool", f"{_fd.name}GymnasiumEnvPool"]
