"""gymnasium adaptor (mirror of envpool/python/gymnasium_envpool.py)."""

from __future__ import annotations

import warnings
from abc import ABCMeta
from typing import Any

import numpy as np

from ._compat import GymEnvBase, spaces
from .data import gymnasium_structure
from .envpool import EnvPoolMixin
from .utils import check_key_duplication


def _env_ids_from_reset_options(options: dict[str, Any] | None,
                                num_envs: int) -> np.ndarray | None:
    """`reset(options={"reset_mask": bool[num_envs]})` -> the env ids to reset, or None for "all envs".

    Contract of envpool/python/gymnasium_envpool.py:51-74 (the only option EnvPool understands is
    `reset_mask`; anything else, a mask of the wrong shape or an all-False mask is a ValueError with the
    reference's wording, which callers match on)."""
    mask = None
    for key, value in (options or {}).items():
        if key != "reset_mask":
            bad = sorted(k for k in options if k != "reset_mask")
            raise ValueError(f"Unsupported Gymnasium reset options for EnvPool: {bad}")
        mask = value
    if mask is None:
        return None
    selected = np.asarray(mask, dtype=np.bool_)
    if selected.shape != (num_envs,):
        raise ValueError(f"reset_mask must have shape ({num_envs},), got {selected.shape}")
    ids = np.nonzero(selected)[0].astype(np.int32)
    if ids.size == 0:
        raise ValueError("reset_mask must select at least one environment.")
    return ids


class GymnasiumEnvPoolMixin:
    """Special treatment for the gymnasium API."""

    metadata = {"render_modes": ["rgb_array", "human"]}

    @property
    def num_envs(self) -> int:
        return int(self.config["num_envs"])

    @property
    def is_vector_env(self) -> bool:
        return True

    @property
    def single_observation_space(self) -> Any:
        return self.observation_space

    @property
    def single_action_space(self) -> Any:
        return self.action_space

    @property
    def observation_space(self) -> Any:
        if not hasattr(self, "_gym_observation_space"):
            self._gym_observation_space = self.spec.gymnasium_observation_space
        return self._gym_observation_space

    @property
    def action_space(self) -> Any:
        if not hasattr(self, "_gym_action_space"):
            self._gym_action_space = self.spec.gymnasium_action_space
        return self._gym_action_space

    @property
    def render_mode(self) -> str | None:
        return getattr(self, "_render_mode", None)

    def reset(self, env_id: np.ndarray | None = None, *,
              seed: int | list[int] | None = None,
              options: dict[str, Any] | None = None) -> Any:
        if seed is not None:
            warnings.warn(
                "EnvPool seeds are fixed when the environment is created. "
                "reset(seed=...) is ignored; pass seed to envpool.make instead.",
                stacklevel=2,
            )
        option_env_id = _env_ids_from_reset_options(options, self.config["num_envs"])
        if env_id is not None and option_env_id is not None:
            raise ValueError("Pass either env_id or options['reset_mask'], not both.")
        if option_env_id is not None:
            # the mask is over this pool's envs; ids are global (env_id_offset extension)
            env_id = option_env_id + np.int32(self.config.get("env_id_offset", 0))
        return super().reset(env_id)

    def close(self, **kwargs: Any) -> None:
        del kwargs
        return super().close()


class GymnasiumEnvPoolMeta(ABCMeta):
    """Builds the gymnasium-flavoured pool class (gymnasium_envpool.py:161-239)."""

    def __new__(cls: Any, name: str, parents: tuple, attrs: dict) -> Any:
        base = parents[0]

        def _xla(self: Any) -> None:
            raise RuntimeError(
                "XLA is unavailable. To enable XLA please install a compatible jax."
            )

        attrs["xla"] = _xla
        parents = (base, GymnasiumEnvPoolMixin, EnvPoolMixin, GymEnvBase)
        state_keys = base._state_keys
        action_keys = base._action_keys
        check_key_duplication(name, "state", state_keys)
        check_key_duplication(name, "action", action_keys)
        build_tree = gymnasium_structure(state_keys)

        def _to_gymnasium(self: Any, state_values: list[np.ndarray], reset: bool,
                          return_info: bool) -> Any:
            state = build_tree(state_values)
            info = state["info"]
            info["elapsed_step"] = state["elapsed_step"]
            obs = state["obs"]
            if not isinstance(self.observation_space, spaces.Dict):
                while isinstance(obs, dict) and len(obs) == 1:
                    obs = next(iter(obs.values()))
            if reset:
                return obs, info
            done = state["done"]
            trunc = state["trunc"]
            terminated = done & ~trunc
            return obs, state["reward"], terminated, trunc, info

        attrs["_to"] = _to_gymnasium
        subcls = super().__new__(cls, name, parents, attrs)

        def init(self: Any, spec: Any) -> None:
            base.__init__(self, spec)
            self.spec = spec

        setattr(subcls, "__init__", init)  # noqa: B010
        return subcls
