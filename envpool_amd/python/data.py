"""Spec -> space transforms and flat-key -> tree conversions.

Host-side mirror of envpool/python/data.py of the reference (same function
names and results); the optree dependency is replaced by the two small
structure builders at the bottom, which produce exactly the trees
`gym_structure` / `dm_structure` + `optree.tree_unflatten` would.
"""

from __future__ import annotations

import keyword
import re
from collections import namedtuple
from typing import Any, Callable

import numpy as np

from ._compat import dm_specs, spaces
from .protocol import ArraySpec

ACTION_THRESHOLD = 2**20


def _maybe_scalar_int(value: Any) -> int | None:
    arr = np.asarray(value)
    if arr.size != 1:
        return None
    scalar = arr.item()
    if not np.isfinite(scalar):
        return None
    integer = int(scalar)
    if not np.isclose(scalar, integer):
        return None
    return integer


def _maybe_discrete_range(spec: ArraySpec, spec_type: str) -> tuple[int, int] | None:
    # envpool/python/data.py:46-62
    if np.prod(np.abs(spec.shape)) != 1:
        return None
    minimum = _maybe_scalar_int(spec.minimum)
    maximum = _maybe_scalar_int(spec.maximum)
    if minimum is None or maximum is None or maximum >= ACTION_THRESHOLD:
        return None
    if spec_type == "act":
        if not (spec.is_discrete or np.issubdtype(spec.dtype, np.integer)):
            return None
    elif not np.issubdtype(spec.dtype, np.integer):
        return None
    return minimum, maximum - minimum + 1


def to_nested_dict(flatten_dict: dict[str, Any], generator: type = dict) -> dict[str, Any]:
    """{"a.b": 1, "a.c": 2} -> {"a": {"b": 1, "c": 2}} (data.py:65-93)."""
    ret: dict[str, Any] = generator()
    for k, v in flatten_dict.items():
        segments = k.split(".")
        ptr = ret
        for s in segments[:-1]:
            if s not in ptr:
                ptr[s] = generator()
            ptr = ptr[s]
        ptr[segments[-1]] = v
    return ret


def _identifier(name: str) -> str:
    ident = re.sub(r"\W", "_", name)
    if not ident or ident[0].isdigit() or keyword.iskeyword(ident):
        ident = f"_{ident}"
    return ident


def to_namedtuple(name: str, hdict: dict) -> tuple:
    """Hierarchical dict -> (nested) namedtuple (data.py:96-117)."""
    field_names = []
    used: dict[str, int] = {}
    for key in hdict.keys():
        field = _identifier(key)
        if field in used:
            used[field] += 1
            field = f"{field}_{used[field]}"
        else:
            used[field] = 0
        field_names.append(field)
    return namedtuple(_identifier(name), field_names)(*[
        to_namedtuple(k, v) if isinstance(v, dict) else v for k, v in hdict.items()
    ])


def dm_spec_transform(name: str, spec: ArraySpec, spec_type: str) -> Any:
    """ArraySpec -> dm_env spec (data.py:120-139)."""
    discrete_range = _maybe_discrete_range(spec, spec_type)
    if discrete_range is not None and discrete_range[0] == 0:
        return dm_specs.DiscreteArray(
            name=name,
            dtype=spec.dtype if np.issubdtype(spec.dtype, np.integer) else np.int32,
            num_values=discrete_range[1],
        )
    return dm_specs.BoundedArray(
        name=name,
        shape=[s for s in spec.shape if s != -1],
        dtype=spec.dtype,
        minimum=spec.minimum,
        maximum=spec.maximum,
    )


def gym_spec_transform(name: str, spec: ArraySpec, spec_type: str) -> Any:
    """ArraySpec -> gymnasium space (data.py:142-157)."""
    discrete_range = _maybe_discrete_range(spec, spec_type)
    if discrete_range is not None:
        start, num_values = discrete_range
        return spaces.Discrete(n=num_values, start=start)
    if np.issubdtype(spec.dtype, np.bool_):
        return spaces.MultiBinary([s for s in spec.shape if s != -1])
    return spaces.Box(
        shape=[s for s in spec.shape if s != -1],
        dtype=spec.dtype,
        low=spec.minimum,
        high=spec.maximum,
    )


gymnasium_spec_transform = gym_spec_transform


# -- structure builders (replace optree flatten/unflatten) -------------------
def gym_structure(keys: list[str]) -> Callable[[list[Any]], dict[str, Any]]:
    """Returns f(state_values) -> nested dict keyed like the reference's
    `gym_structure` tree (data.py:192-204): ':' and '.' both nest."""
    paths = [k.replace(":", ".").split(".") for k in keys]

    def build(values: list[Any]) -> dict[str, Any]:
        root: dict[str, Any] = {}
        for path, v in zip(paths, values):
            ptr = root
            for s in path[:-1]:
                ptr = ptr.setdefault(s, {})
            ptr[path[-1]] = v
        return root

    return build


gymnasium_structure = gym_structure


def dm_structure(root_name: str, keys: list[str]) -> Callable[[list[Any]], tuple]:
    """Returns f(state_values) -> namedtuple tree of the reference's
    `dm_structure` (data.py:160-189): obs:* and info:* merge under `State`."""
    new_keys = []
    for key in keys:
        if key in ["obs", "info"]:
            key = f"obs:{key}"
        key = key.replace("info:", "obs:")
        key = key.replace("obs:", f"{root_name}:")
        new_keys.append(key.replace(":", "."))
    dict_tree = to_nested_dict(dict(zip(new_keys, range(len(new_keys)))))
    template = to_namedtuple(root_name, dict_tree)

    def fill(node: Any, values: list[Any]) -> Any:
        if isinstance(node, tuple):
            return type(node)(*[fill(c, values) for c in node])
        return values[node]

    def build(values: list[Any]) -> tuple:
        return fill(template, values)

    return build
