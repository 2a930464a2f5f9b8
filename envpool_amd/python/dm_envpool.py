"""dm_env adaptor: turns a native `_XxxEnvPool` class into a `dm_env.Environment`
whose `reset/step/recv` hand out `TimeStep(step_type, reward, discount,
observation=State namedtuple)` -- the contract of envpool/python/dm_envpool.py:45-103
(`DMEnvPoolMeta`), checked by tests/test_host_api.py::test_dm_adaptor_contract."""

from __future__ import annotations

from abc import ABCMeta
from typing import Any

import numpy as np

from ._compat import DMEnvBase, TimeStep
from .data import dm_structure
from .envpool import EnvPoolMixin
from .utils import check_key_duplication


def _no_xla(self: Any) -> None:
    raise RuntimeError("XLA is unavailable. To enable XLA please install a compatible jax.")


class DMEnvPoolMixin:
    """dm_env spells the spec accessors as methods; cache what the spec object builds."""

    def _cached(self, slot: str, maker: str) -> Any:
        if slot not in self.__dict__:
            self.__dict__[slot] = getattr(self.spec, maker)()
        return self.__dict__[slot]

    def observation_spec(self) -> tuple:
        return self._cached("_dm_observation_spec", "observation_spec")

    def action_spec(self) -> Any:
        return self._cached("_dm_action_spec", "action_spec")


class _TimeStepBuilder:
    """`_to` hook of EnvPoolMixin: flat state arrays -> dm_env TimeStep."""

    def __init__(self, state_keys: list[str]) -> None:
        self._tree = dm_structure("State", state_keys)

    def __call__(self, pool: Any, state_values: list[np.ndarray], reset: bool,
                 return_info: bool) -> Any:
        state = self._tree(state_values)
        return TimeStep(step_type=state.step_type, reward=state.reward,
                        discount=state.discount, observation=state.State)


class DMEnvPoolMeta(ABCMeta):
    """Metaclass producing the dm_env-flavoured pool class of one env family."""

    def __new__(cls: Any, name: str, parents: tuple, attrs: dict) -> Any:
        native = parents[0]
        for kind in ("state", "action"):
            check_key_duplication(name, kind, getattr(native, f"_{kind}_keys"))
        build = _TimeStepBuilder(native._state_keys)
        attrs.update(xla=_no_xla,
                     _to=lambda self, values, reset, return_info: build(self, values, reset,
                                                                         return_info))
        pool_cls = super().__new__(cls, name, (native, DMEnvPoolMixin, EnvPoolMixin, DMEnvBase),
                                   attrs)

        def __init__(self: Any, spec: Any) -> None:
            native.__init__(self, spec)
            self.spec = spec

        pool_cls.__init__ = __init__
        return pool_cls
