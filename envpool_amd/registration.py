"""Global env registry: make / make_gym / make_dm / make_spec / list_all_envs.

Host-side mirror of envpool/registration.py (same functions, kwargs checks and
errors); pixel/render variants are out of scope, so `from_pixels=True` raises
and render kwargs are only carried as attributes like the reference does.
"""

from __future__ import annotations

import importlib
import os
from collections.abc import Sequence
from typing import Any

import numpy as np

package_base_path = os.path.abspath(os.path.dirname(__file__))
base_path = package_base_path


class EnvRegistry:
    """A collection of available envs."""

    def __init__(self) -> None:
        self.specs: dict[str, tuple[str, str, dict[str, Any]]] = {}
        self.envpools: dict[str, dict[str, tuple[str, str]]] = {}

    def register(self, task_id: str, import_path: str, spec_cls: str, dm_cls: str,
                 gymnasium_cls: str, aliases: Sequence[str] = (), **kwargs: Any) -> None:
        """One task id (and its aliases) -> where its spec class and its two pool classes live, plus the
        config defaults of the id (registration.py:72-92)."""
        kwargs.setdefault("base_path", base_path)
        pools = {"dm": (import_path, dm_cls), "gymnasium": (import_path, gymnasium_cls)}
        for name in (task_id, *aliases):
            assert name not in self.specs
            self.specs[name] = (import_path, spec_cls, dict(kwargs))
            self.envpools[name] = dict(pools)

    _RENDER_MODES = (None, "rgb_array", "human")
    _RENDER_KEYS = ("render_width", "render_height", "render_camera_id")

    @classmethod
    def _extract_make_options(cls, kwargs: dict[str, Any]) -> tuple[bool, dict[str, Any]]:
        """Takes the wrapper-level options out of `kwargs` (registration.py:94-139; render bookkeeping only:
        they become attributes of the env object, nothing is rendered here)."""
        from_pixels = bool(kwargs.pop("from_pixels", False))
        wrapper_kwargs = {}
        for key in ("render_mode", "render_env_id"):
            if key in kwargs:
                wrapper_kwargs[key] = kwargs.pop(key)
        if wrapper_kwargs.get("render_mode") not in cls._RENDER_MODES:
            raise ValueError("render_mode must be one of None, 'rgb_array', or 'human'")
        if from_pixels:
            raise ValueError(
                "from_pixels=True needs the reference's offscreen renderer, which is "
                "outside the batched-step path this engine replaces."
            )
        for key in cls._RENDER_KEYS:
            if key in kwargs:
                wrapper_kwargs[key] = kwargs.pop(key)
        return from_pixels, wrapper_kwargs

    @staticmethod
    def _apply_wrapper_kwargs(env: Any, wrapper_kwargs: dict[str, Any]) -> Any:
        for key, value in wrapper_kwargs.items():
            setattr(env, f"_{key}", value)
        return env

    def _make_env_spec(self, task_id: str, **make_kwargs: Any) -> Any:
        """The id's registered defaults overridden by the caller's kwargs, checked like the reference checks them
        (registration.py:187-234: AssertionError for out-of-range values), then the spec class's gen_config."""
        import_path, spec_cls, defaults = self.specs[task_id]
        kwargs = dict(defaults)
        kwargs.update(make_kwargs)
        num_envs = kwargs.get("num_envs", 1)
        if "seed" in kwargs:
            seed = kwargs["seed"]
            if self._is_env_seed_sequence(seed):  # a list of seeds is the per-env form (reference issue 214)
                assert "env_seed" not in kwargs, (
                    "Pass either `seed` as an int or seed list, or `env_seed`, but not both."
                )
                kwargs["env_seed"], kwargs["seed"] = seed, 0
            else:
                self._assert_int32_seed(seed)
        if "env_seed" in kwargs:
            kwargs["env_seed"] = self._normalize_env_seed(kwargs["env_seed"], num_envs)
        assert kwargs.get("num_envs", 1) >= 1
        if "batch_size" in kwargs:
            assert 0 <= kwargs["batch_size"] <= kwargs["num_envs"]
        assert kwargs.get("max_num_players", 1) >= 1
        spec_type = getattr(importlib.import_module(import_path), spec_cls)
        return spec_type(spec_type.gen_config(**kwargs))

    def make(self, task_id: str, env_type: str, **kwargs: Any) -> Any:
        """registration.py:250-281."""
        _, wrapper_kwargs = self._extract_make_options(kwargs)
        if not kwargs.setdefault("gym_reset_return_info", True):
            raise ValueError(
                "EnvPool's gym API now follows gymnasium reset semantics and "
                "always returns an info dictionary after resets."
            )
        assert task_id in self.specs, (
            f"{task_id} is not supported, `envpool.list_all_envs()` may help."
        )
        assert env_type in ["dm", "gymnasium"]
        spec = self._make_env_spec(task_id, **kwargs)
        import_path, envpool_cls = self.envpools[task_id][env_type]
        env = getattr(importlib.import_module(import_path), envpool_cls)(spec)
        return self._apply_wrapper_kwargs(env, wrapper_kwargs)

    def make_dm(self, task_id: str, **kwargs: Any) -> Any:
        return self.make(task_id, "dm", **kwargs)

    def make_gymnasium(self, task_id: str, **kwargs: Any) -> Any:
        return self.make(task_id, "gymnasium", **kwargs)

    def make_spec(self, task_id: str, **make_kwargs: Any) -> Any:
        self._extract_make_options(make_kwargs)
        return self._make_env_spec(task_id, **make_kwargs)

    @staticmethod
    def _assert_int32_seed(seed: Any) -> None:
        INT_MAX = 2**31
        assert -INT_MAX <= seed < INT_MAX, f"Seed should be in range of int32, got {seed}"

    @staticmethod
    def _is_env_seed_sequence(seed: Any) -> bool:
        return (isinstance(seed, Sequence) and not isinstance(seed, (str, bytes))) or \
            isinstance(seed, np.ndarray)

    def _normalize_env_seed(self, seed: Any, num_envs: int) -> list[int]:
        """per-env seeds as a list of num_envs Python ints, each an int32 (registration.py:313-330)"""
        if isinstance(seed, np.ndarray):
            assert seed.ndim == 1, f"`seed` as an array must be 1-dimensional, got shape {seed.shape}"
        seeds = [int(x) for x in (seed.tolist() if isinstance(seed, np.ndarray) else seed)]
        assert len(seeds) == num_envs, (
            "When `seed` is a sequence, its length must match `num_envs`, "
            f"got len(seed) = {len(seeds)} and num_envs = {num_envs}"
        )
        for x in seeds:
            self._assert_int32_seed(x)
        return seeds

    def list_all_envs(self) -> list[str]:
        return list(self.specs.keys())


registry = EnvRegistry()
register = registry.register


def make(task_id: str, env_type: str, **kwargs: Any) -> Any:
    """Make an EnvPool (registration.py:364-378)."""
    if env_type == "dm":
        return registry.make(task_id, "dm", **kwargs)
    if env_type in ("gym", "gymnasium"):
        return registry.make(task_id, "gymnasium", **kwargs)
    raise AssertionError("env_type should be one of 'dm', 'gym', or 'gymnasium'.")


def make_dm(task_id: str, **kwargs: Any) -> Any:
    return registry.make_dm(task_id, **kwargs)


def make_gym(task_id: str, **kwargs: Any) -> Any:
    return make_gymnasium(task_id, **kwargs)


def make_gymnasium(task_id: str, **kwargs: Any) -> Any:
    return registry.make_gymnasium(task_id, **kwargs)


def make_spec(task_id: str, **kwargs: Any) -> Any:
    return registry.make_spec(task_id, **kwargs)


def list_all_envs() -> list[str]:
    return registry.list_all_envs()
