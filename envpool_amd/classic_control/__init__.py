"""Classic control envs (mirror of envpool/classic_control/__init__.py).

Spec tables restate `XxxEnvFns::{DefaultConfig,StateSpec,ActionSpec}`:
  cartpole.h:31-46, pendulum.h:31-44, mountain_car.h:31-46,
  mountain_car_continuous.h:31-46, acrobot.h:31-48.
"""

import math

import numpy as np

from envpool_amd.core.binding import FamilyDef, make_native_classes, spec
from envpool_amd.python.api import py_env

_inf = float("inf")


def _f32(vals):
    return [float(np.float32(v)) for v in vals]


_CartPole = FamilyDef(
    name="CartPole", native="CartPole",
    default_config=[("reward_threshold", 195.0)],
    state_spec=lambda c: [("obs", spec(np.float32, [4], None, (
        _f32([-4.8, -_inf, -math.pi / 7.5, -_inf]), _f32([4.8, _inf, math.pi / 7.5, _inf]))))],
    action_spec=lambda c: [("action", spec(np.int32, [-1], (0, 1)))],
)
_Pendulum = FamilyDef(
    name="Pendulum", native="Pendulum",
    default_config=[("version", 0)],
    state_spec=lambda c: [("obs", spec(np.float32, [3], None, (
        _f32([-1.0, -1.0, -8.0]), _f32([1.0, 1.0, 8.0]))))],
    action_spec=lambda c: [("action", spec(np.float32, [-1, 1], (-2.0, 2.0)))],
    native_params=lambda c: {"version": c["version"]},
)
_MountainCar = FamilyDef(
    name="MountainCar", native="MountainCar",
    default_config=[("reward_threshold", -110.0)],
    state_spec=lambda c: [("obs", spec(np.float32, [2], None, (
        _f32([-1.2, -0.07]), _f32([0.6, 0.07]))))],
    action_spec=lambda c: [("action", spec(np.int32, [-1], (0, 2)))],
)
_MountainCarContinuous = FamilyDef(
    name="MountainCarContinuous", native="MountainCarContinuous",
    default_config=[("reward_threshold", 90.0)],
    state_spec=lambda c: [("obs", spec(np.float32, [2], None, (
        _f32([-1.2, -0.07]), _f32([0.6, 0.07]))))],
    action_spec=lambda c: [("action", spec(np.float32, [-1, 1], (-1.0, 1.0)))],
)
_Acrobot = FamilyDef(
    name="Acrobot", native="Acrobot",
    default_config=[("reward_threshold", -100.0)],
    state_spec=lambda c: [
        ("obs", spec(np.float32, [6], None, (
            _f32([-1.0, -1.0, -1.0, -1.0, -4 * math.pi, -9 * math.pi]),
            _f32([1.0, 1.0, 1.0, 1.0, 4 * math.pi, 9 * math.pi])))),
        ("info:state", spec(np.float32, [2])),
    ],
    action_spec=lambda c: [("action", spec(np.int32, [-1], (0, 2)))],
)

_CartPoleEnvSpec, _CartPoleEnvPool = make_native_classes(_CartPole)
_PendulumEnvSpec, _PendulumEnvPool = make_native_classes(_Pendulum)
_MountainCarEnvSpec, _MountainCarEnvPool = make_native_classes(_MountainCar)
_MountainCarContinuousEnvSpec, _MountainCarContinuousEnvPool = make_native_classes(
    _MountainCarContinuous)
_AcrobotEnvSpec, _AcrobotEnvPool = make_native_classes(_Acrobot)

CartPoleEnvSpec, CartPoleDMEnvPool, CartPoleGymnasiumEnvPool = py_env(
    _CartPoleEnvSpec, _CartPoleEnvPool)
PendulumEnvSpec, PendulumDMEnvPool, PendulumGymnasiumEnvPool = py_env(
    _PendulumEnvSpec, _PendulumEnvPool)
MountainCarEnvSpec, MountainCarDMEnvPool, MountainCarGymnasiumEnvPool = py_env(
    _MountainCarEnvSpec, _MountainCarEnvPool)
(MountainCarContinuousEnvSpec, MountainCarContinuousDMEnvPool,
 MountainCarContinuousGymnasiumEnvPool) = py_env(
    _MountainCarContinuousEnvSpec, _MountainCarContinuousEnvPool)
AcrobotEnvSpec, AcrobotDMEnvPool, AcrobotGymnasiumEnvPool = py_env(
    _AcrobotEnvSpec, _AcrobotEnvPool)

__all__ = [
    "CartPoleEnvSpec", "CartPoleDMEnvPool", "CartPoleGymnasiumEnvPool",
    "PendulumEnvSpec", "PendulumDMEnvPool", "PendulumGymnasiumEnvPool",
    "MountainCarEnvSpec", "MountainCarDMEnvPool", "MountainCarGymnasiumEnvPool",
    "MountainCarContinuousEnvSpec", "MountainCarContinuousDMEnvPool",
    "MountainCarContinuousGymnasiumEnvPool",
    "AcrobotEnvSpec", "AcrobotDMEnvPool", "AcrobotGymnasiumEnvPool",
]
