"""dm_env adaptor (mirror of envpool/python/dm_envpool.py)."""

from __future__ import annotations

from abc import ABCMeta
from typing import Any

import numpy as np

from ._compat import DMEnvBase, TimeStep
from .data import dm_structure
from .envpool import EnvPoolMixin
from .utils import check_key_duplication


class DMEnvPoolMixin:
    """Special treatment for the dm_env API."""

    def observation_spec(self) -> tuple:
        if not hasattr(self, "_dm_observation_spec"):
            self._dm_observation_spec = self.spec.observation_spec()
        return self._dm_observation_spec

    def action_spec(self) -> Any:
        if not hasattr(self, "_dm_action_spec"):
            self._dm_action_spec = self.spec.action_spec()
        return self._dm_action_spec


class DMEnvPoolMeta(ABCMeta):
    """Builds the dm_env-flavoured pool class (dm_envpool.py:45-103)."""

    def __new__(cls: Any, name: str, parents: tuple, attrs: dict) -> Any:
        base = parents[0]

        def _xla(self: Any) -> None:
            raise RuntimeError(
                "XLA is unavailable. To enable XLA please install a compatible jax."
            )

        attrs["xla"] = _xla
        parents = (base, DMEnvPoolMixin, EnvPoolMixin, DMEnvBase)
        state_keys = base._state_keys
        action_keys = base._action_keys
        check_key_duplication(name, "state", state_keys)
        check_key_duplication(name, "action", action_keys)
        build_tree = dm_structure("State", state_keys)

        def _to_dm(self: Any, state_values: list[np.ndarray], reset: bool,
                   return_info: bool) -> Any:
            state = build_tree(state_values)
            return TimeStep(
                step_type=state.step_type,
                observation=state.State,
                reward=state.reward,
                discount=state.discount,
            )

        attrs["_to"] = _to_dm
        subcls = super().__new__(cls, name, parents, attrs)

        def init(self: Any, spec: Any) -> None:
            base.__init__(self, spec)
            self.spec = spec

        setattr(subcls, "__init__", init)  # noqa: B010
        return subcls
