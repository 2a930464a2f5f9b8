"""EnvSpecMixin / EnvSpecMeta: python-side view of a native spec class.

Host-side mirror of envpool/python/env_spec.py (same properties and results).
"""

from __future__ import annotations

import pprint
from abc import ABC, ABCMeta
from collections import namedtuple
from typing import Any, NamedTuple

from ._compat import spaces
from .data import dm_spec_transform, gym_spec_transform, to_namedtuple, to_nested_dict
from .protocol import ArraySpec
from .utils import check_key_duplication


class EnvSpecMixin(ABC):
    """Mixin class for EnvSpec, exposed to EnvSpecMeta."""

    gen_config: type

    @property
    def config(self) -> NamedTuple:
        return self.gen_config(*self._config_values)

    @property
    def reward_threshold(self) -> float | None:
        try:
            return self.config.reward_threshold
        except AttributeError:
            return None

    @property
    def state_array_spec(self) -> dict[str, Any]:
        return dict(zip(self._state_keys, [ArraySpec(*s) for s in self._state_spec]))

    @property
    def action_array_spec(self) -> dict[str, Any]:
        return dict(zip(self._action_keys, [ArraySpec(*s) for s in self._action_spec]))

    def observation_spec(self) -> tuple:
        """dm_env observation spec: obs:* and info:* keys (env_spec.py:74-90)."""
        spec = {
            k.replace("obs:", "").replace("info:", ""): dm_spec_transform(
                k.replace(":", ".").split(".")[-1], v, "obs"
            )
            for k, v in self.state_array_spec.items()
            if k.startswith(("obs", "info"))
        }
        return to_namedtuple("State", to_nested_dict(spec))

    def action_spec(self) -> Any:
        """dm_env action spec (env_spec.py:92-117)."""
        spec = self.action_array_spec
        if len(spec) == 3:
            spec.pop("env_id")
            spec.pop("players.env_id")
            return dm_spec_transform(list(spec.keys())[0], list(spec.values())[0], "act")
        spec = {k: dm_spec_transform(k.split(".")[-1], v, "act") for k, v in spec.items()}
        return to_namedtuple("Action", to_nested_dict(spec))

    @property
    def observation_space(self) -> Any:
        """gym observation space: keys starting with obs (env_spec.py:119-141)."""
        spec = {
            k.replace("obs:", "").replace(":", "."): gym_spec_transform(
                k.replace(":", ".").split(".")[-1], v, "obs"
            )
            for k, v in self.state_array_spec.items()
            if k.startswith("obs")
        }
        if len(spec) == 1:
            return list(spec.values())[0]
        return to_nested_dict(spec, spaces.Dict)

    @property
    def action_space(self) -> Any:
        """gym action space (env_spec.py:143-169)."""
        spec = self.action_array_spec
        if len(spec) == 3:
            spec.pop("env_id")
            spec.pop("players.env_id")
            return gym_spec_transform(list(spec.keys())[0], list(spec.values())[0], "act")
        spec = {k: gym_spec_transform(k.split(".")[-1], v, "act") for k, v in spec.items()}
        return to_nested_dict(spec, spaces.Dict)

    @property
    def gymnasium_observation_space(self) -> Any:
        return self.observation_space

    @property
    def gymnasium_action_space(self) -> Any:
        return self.action_space

    def __repr__(self) -> str:
        config_info = pprint.pformat(self.config)[6:]
        return f"{self.__class__.__name__}{config_info}"


class EnvSpecMeta(ABCMeta):
    """Checks keys and attaches the `gen_config` namedtuple (env_spec.py:205-222)."""

    def __new__(cls: Any, name: str, parents: tuple, attrs: dict) -> Any:
        base = parents[0]
        parents = (base, EnvSpecMixin)
        raw_config_keys = base._config_keys
        check_key_duplication(name, "config", raw_config_keys)
        config_keys = [s.replace(".", "_") for s in raw_config_keys]
        defaults: tuple = base._default_config_values
        attrs["gen_config"] = namedtuple("Config", config_keys, defaults=defaults)
        return super().__new__(cls, name, parents, attrs)
