"""EnvPoolMixin: send / recv / step / reset / async_reset on top of the native
pool's `_send / _recv / _reset`.

Host-side mirror of envpool/python/envpool.py:61-384 (same method names,
argument meaning and error behaviour).  Rendering stays out of scope: `render`
keeps the reference's checks and then surfaces the pool's RuntimeError.
"""

from __future__ import annotations

import pprint
import warnings
from abc import ABC
from typing import Any

import numpy as np


def _normalize_env_id(env_id: Any) -> Any:
    """env ids as an int32 array of at least one dimension (envpool.py:38-48).  Array-likes with their own `astype`
    (device arrays) keep their type; everything else goes through numpy."""
    if hasattr(env_id, "astype"):
        # numpy: no copy when the dtype already matches; other array types: their own conversion
        ids = env_id.astype(np.int32, copy=False) if isinstance(env_id, np.ndarray) else env_id.astype(np.int32)
    else:
        ids = np.asarray(env_id, dtype=np.int32)
    return ids.reshape(1) if getattr(ids, "ndim", 0) == 0 else ids


def _flatten_action_dict(action: dict, prefix: tuple = ()) -> dict[str, Any]:
    """{"a": {"b": x}} -> {"a.b": x} (the reference uses optree paths)."""
    out: dict[str, Any] = {}
    for k, v in action.items():
        if isinstance(v, dict):
            out.update(_flatten_action_dict(v, prefix + (k,)))
        else:
            out[".".join(prefix + (k,))] = v
    return out


class EnvPoolMixin(ABC):
    """Mixin class for EnvPool, exposed to the gymnasium / dm metaclasses."""

    def _check_action(self, actions: list[np.ndarray]) -> None:
        """dtype and per-row shape of every action array against the spec -- on the FIRST send only
        (envpool.py:151-172); the messages are the reference's."""
        if getattr(self, "_check_action_finished", False):
            return
        self._check_action_finished = True
        specs = self.spec.action_array_spec
        for arr, (name, want) in zip(actions, specs.items()):
            if arr.dtype != want.dtype:
                raise RuntimeError(f'Expected dtype {want.dtype} with action "{name}", got {arr.dtype}')
            shape = tuple(want.shape)
            per_player = len(shape) > 0 and shape[0] == -1  # leading -1: one row per player
            if per_player:
                ok, shown = arr.shape[1:] == shape[1:], shape
            else:
                ok, shown = arr.ndim > 0 and arr.shape[1:] == shape, ("num_env", *shape)
            if not ok:
                raise RuntimeError(f'Expected shape {shown} with action "{name}", got {arr.shape}')

    def _from(self, action: dict[str, Any] | np.ndarray,
              env_id: np.ndarray | None = None) -> list[np.ndarray]:
        """Convert an action into the native list (envpool.py:174-208)."""
        if isinstance(action, dict):
            adict = _flatten_action_dict(action)
        else:
            # a bare array is the LAST action key (the other two are the env ids); its dtype comes from the spec
            if not hasattr(self, "_last_action_name"):
                self._last_action_name = self._spec._action_keys[-1]
                self._last_action_type = self._spec._action_spec[-1][0]
            if isinstance(action, np.ndarray):
                # (the reference copies here; the pool stages the rows before `send` returns, so an
                # array that already has the dtype and layout can be passed through)
                action = action.astype(self._last_action_type, order="C", copy=False)
            adict = {self._last_action_name: action}
        if env_id is not None:
            adict["env_id"] = env_id.astype(np.int32, copy=False)
        else:
            adict.setdefault("env_id", self.all_env_ids)
        if "players.env_id" not in adict:
            # all hot-path envs are single player: players.env_id == env_id
            adict["players.env_id"] = _normalize_env_id(adict["env_id"])
        if not hasattr(self, "_action_names"):
            self._action_names = self._spec._action_keys
        return [adict[name] for name in self._action_names]

    def __len__(self) -> int:
        return self.config["num_envs"]

    @property
    def all_env_ids(self) -> np.ndarray:
        if not hasattr(self, "_all_env_ids"):
            # `env_id_offset` (extension): this pool is one shard of a bigger pool and
            # its env ids are the global ones [offset, offset + num_envs)
            off = int(self.config.get("env_id_offset", 0))
            self._all_env_ids = np.arange(off, off + self.config["num_envs"], dtype=np.int32)
        return self._all_env_ids

    @property
    def is_async(self) -> bool:
        return (self.config["batch_size"] > 0
                and self.config["num_envs"] != self.config["batch_size"])

    def seed(self, seed: int | list[int] | None = None) -> None:
        warnings.warn(
            "The `seed` function in envpool is abandoned. "
            "You can set seed by envpool.make(..., seed=seed) instead.",
            stacklevel=2,
        )

    def render(self, env_ids: Any = None, camera_id: int | None = None) -> Any:
        render_mode = getattr(self, "_render_mode", None)
        if render_mode not in {"rgb_array", "human"}:
            raise RuntimeError(
                "render_mode must be set to 'rgb_array' or 'human' when creating this env"
            )
        if env_ids is None:
            env_ids = [int(getattr(self, "_render_env_id", 0))]
        ids = np.atleast_1d(np.asarray(env_ids, dtype=np.int32))
        return self._render(
            ids,
            int(getattr(self, "_render_width", 0)),
            int(getattr(self, "_render_height", 0)),
            int(getattr(self, "_render_camera_id", -1) if camera_id is None else camera_id),
        )

    def send(self, action: dict[str, Any] | np.ndarray,
             env_id: np.ndarray | None = None) -> None:
        converted_action = self._from(action, env_id)
        self._check_action(converted_action)
        self._send(converted_action)

    def recv(self, reset: bool = False, return_info: bool = True) -> Any:
        state_list = self._recv()
        return self._to(state_list, reset, return_info)

    def async_reset(self) -> None:
        self._reset(self.all_env_ids)

    def step(self, action: dict[str, Any] | np.ndarray,
             env_id: np.ndarray | None = None) -> Any:
        self.send(action, env_id)
        return self.recv(reset=False, return_info=True)

    def reset(self, env_id: np.ndarray | None = None) -> Any:
        if env_id is None:
            env_id = self.all_env_ids
        self._reset(env_id)
        return self.recv(reset=True, return_info=self.config["gym_reset_return_info"])

    def close(self) -> None:
        close = getattr(super(), "close", None)
        if close is not None:
            close()

    @property
    def config(self) -> dict[str, Any]:
        return dict(zip(self._spec._config_keys, self._spec._config_values))

    def __repr__(self) -> str:
        config_str = ", ".join(f"{k}={pprint.pformat(v)}" for k, v in self.config.items())
        return f"{self.__class__.__name__}({config_str})"

    __str__ = __repr__
