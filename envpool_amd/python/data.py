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


def _as_whole_number(value: Any) -> int | None:
    """`value` as a Python int when it is ONE finite number with no fractional part (up to np.isclose), else None:
    the reference's bound test for "this spec is a discrete range" (envpool/python/data.py:32-42)."""
    flat = np.ravel(np.asarray(value))
    if flat.size != 1:
        return None
    x = flat[0].item()
    if not np.isfinite(x) or not np.isclose(x, int(x)):
        return None
    return int(x)


def _maybe_discrete_range(spec: ArraySpec, spec_type: str) -> tuple[int, int] | None:
    """(start, number of values) when `spec` describes a scalar integer range, else None (data.py:45-62).
    An action spec also counts when it is flagged discrete; a state spec must have an integer dtype."""
    integral = np.issubdtype(spec.dtype, np.integer)
    if spec_type == "act":
        integral = integral or bool(spec.is_discrete)
    scalar = int(np.prod(np.abs(spec.shape))) == 1
    if not (integral and scalar):
        return None
    lo, hi = _as_whole_number(spec.minimum), _as_whole_number(spec.maximum)
    if lo is None or hi is None or hi >= ACTION_THRESHOLD:
        return None
    return lo, hi - lo + 1


def _static_shape(spec: ArraySpec) -> list[int]:
    """the spec's shape without the per-player placeholder (-1)"""
    return [dim for dim in spec.shape if dim != -1]


def to_nested_dict(flatten_dict: dict[str, Any], generator: type = dict) -> dict[str, Any]:
    """{"a.b": 1, "a.c": 2} -> {"a": {"b": 1, "c": 2}} (data.py:65-93)."""
    tree: dict[str, Any] = generator()
    for dotted, leaf in flatten_dict.items():
        *parents, last = dotted.split(".")
        node = tree
        for name in parents:
            if name not in node:
                node[name] = generator()
            node = node[name]
        node[last] = leaf
    return tree


def _identifier(name: str) -> str:
    """a legal namedtuple type / field name for `name` (data.py:98-100, 104-106)"""
    ident = re.sub(r"\W", "_", name)
    needs_prefix = ident == "" or ident[0].isdigit() or keyword.iskeyword(ident)
    return "_" + ident if needs_prefix else ident


def to_namedtuple(name: str, hdict: dict) -> tuple:
    """Hierarchical dict -> (nested) namedtuple (data.py:96-117); a field name that repeats after
    sanitising gets the suffix _1, _2, ..."""
    seen: dict[str, int] = {}
    fields, values = [], []
    for key, child in hdict.items():
        field = _identifier(key)
        repeats = seen.get(field)
        seen[field] = 0 if repeats is None else repeats + 1
        fields.append(field if repeats is None else f"{field}_{repeats + 1}")
        values.append(to_namedtuple(key, child) if isinstance(child, dict) else child)
    return namedtuple(_identifier(name), fields)(*values)


def dm_spec_transform(name: str, spec: ArraySpec, spec_type: str) -> Any:
    """ArraySpec -> dm_env spec (data.py:120-139): a zero-based discrete range becomes a DiscreteArray
    (dm_env has no other kind), everything else a BoundedArray."""
    rng = _maybe_discrete_range(spec, spec_type)
    if rng is None or rng[0] != 0:
        return dm_specs.BoundedArray(name=name, shape=_static_shape(spec), dtype=spec.dtype,
                                     minimum=spec.minimum, maximum=spec.maximum)
    int_dtype = spec.dtype if np.issubdtype(spec.dtype, np.integer) else np.int32
    return dm_specs.DiscreteArray(name=name, dtype=int_dtype, num_values=rng[1])


def gym_spec_transform(name: str, spec: ArraySpec, spec_type: str) -> Any:
    """ArraySpec -> gymnasium space (data.py:142-157): Discrete (any start), MultiBinary for bool, else Box."""
    rng = _maybe_discrete_range(spec, spec_type)
    if rng is not None:
        return spaces.Discrete(n=rng[1], start=rng[0])
    if np.issubdtype(spec.dtype, np.bool_):
        return spaces.MultiBinary(_static_shape(spec))
    return spaces.Box(low=spec.minimum, high=spec.maximum, shape=_static_shape(spec), dtype=spec.dtype)


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
    def dotted(key: str) -> str:
        # a bare "obs" / "info" key is a leaf of the root; info:* joins obs:* under the root's name
        if key in ("obs", "info"):
            key = "obs:" + key
        return key.replace("info:", "obs:").replace("obs:", root_name + ":").replace(":", ".")

    index_of = {dotted(key): i for i, key in enumerate(keys)}
    template = to_namedtuple(root_name, to_nested_dict(index_of))

    def fill(node: Any, values: list[Any]) -> Any:
        if isinstance(node, tuple):
            return type(node)(*[fill(c, values) for c in node])
        return values[node]

    def build(values: list[Any]) -> tuple:
        return fill(template, values)

    return build
