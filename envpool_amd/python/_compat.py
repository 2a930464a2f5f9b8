"""Optional third-party interfaces of the reference's Python layer.

The reference imports gymnasium, dm_env and optree unconditionally
(envpool/python/{data,envpool,gymnasium_envpool,dm_envpool}.py).  None of them
is installed in the build image, so each is used when importable and otherwise
replaced by the minimal stand-in below that offers the attributes the adaptors
(and RL code written against envpool) actually touch.  Nothing here is on the
step path.
"""

from __future__ import annotations

import enum
from collections import namedtuple
from typing import Any

import numpy as np

try:  # pragma: no cover - depends on the environment
    import gymnasium  # type: ignore

    HAVE_GYMNASIUM = True
except ImportError:  # pragma: no cover
    gymnasium = None
    HAVE_GYMNASIUM = False

try:  # pragma: no cover
    import dm_env  # type: ignore

    HAVE_DM_ENV = True
except ImportError:  # pragma: no cover
    dm_env = None
    HAVE_DM_ENV = False


# --------------------------------------------------------------------------
# gymnasium stand-ins
# --------------------------------------------------------------------------
class _Space:
    shape: tuple = ()
    dtype: Any = None

    def contains(self, x: Any) -> bool:  # pragma: no cover - trivial
        raise NotImplementedError

    def __contains__(self, x: Any) -> bool:
        return self.contains(x)


class _Box(_Space):
    def __init__(self, low: Any, high: Any, shape: Any = None, dtype: Any = np.float32):
        self.dtype = np.dtype(dtype)
        shape = tuple(int(s) for s in (shape if shape is not None else np.shape(low)))
        self.shape = shape
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), shape).copy()
        self._rng = np.random.default_rng()

    def sample(self) -> np.ndarray:
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return self._rng.uniform(lo, hi).astype(self.dtype)

    def contains(self, x: Any) -> bool:
        x = np.asarray(x)
        return bool(x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self) -> str:
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    def __eq__(self, o: Any) -> bool:
        return (isinstance(o, _Box) and self.shape == o.shape and
                np.array_equal(self.low, o.low) and np.array_equal(self.high, o.high))


class _Discrete(_Space):
    def __init__(self, n: int, start: int = 0):
        self.n = int(n)
        self.start = int(start)
        self.shape = ()
        self.dtype = np.dtype(np.int64)
        self._rng = np.random.default_rng()

    def sample(self) -> int:
        return int(self.start + self._rng.integers(self.n))

    def contains(self, x: Any) -> bool:
        try:
            v = int(x)
        except (TypeError, ValueError):
            return False
        return self.start <= v < self.start + self.n

    def __repr__(self) -> str:
        return f"Discrete({self.n})" if self.start == 0 else f"Discrete({self.n}, start={self.start})"

    def __eq__(self, o: Any) -> bool:
        return isinstance(o, _Discrete) and (self.n, self.start) == (o.n, o.start)


class _MultiBinary(_Space):
    def __init__(self, n: Any):
        self.n = n
        self.shape = tuple(n) if isinstance(n, (list, tuple)) else (int(n),)
        self.dtype = np.dtype(np.int8)


class _DictSpace(_Space, dict):
    def __init__(self, spaces: dict | None = None):
        dict.__init__(self, spaces or {})

    @property
    def spaces(self) -> dict:
        return self


class _Env:
    """Stand-in for gymnasium.Env (only used as a base class)."""

    metadata: dict = {}


class _SpacesModule:
    Space = _Space
    Box = _Box
    Discrete = _Discrete
    MultiBinary = _MultiBinary
    Dict = _DictSpace


if HAVE_GYMNASIUM:  # pragma: no cover
    spaces = gymnasium.spaces
    GymEnvBase = gymnasium.Env
else:
    spaces = _SpacesModule
    GymEnvBase = _Env


# --------------------------------------------------------------------------
# dm_env stand-ins
# --------------------------------------------------------------------------
class _StepType(enum.IntEnum):
    FIRST = 0
    MID = 1
    LAST = 2


class _TimeStep(namedtuple("TimeStep", ["step_type", "reward", "discount", "observation"])):
    __slots__ = ()

    def first(self) -> Any:
        return self.step_type == _StepType.FIRST

    def mid(self) -> Any:
        return self.step_type == _StepType.MID

    def last(self) -> Any:
        return self.step_type == _StepType.LAST


class _ArraySpecDM:
    def __init__(self, shape: Any, dtype: Any, name: str | None = None):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.name = name

    def __repr__(self) -> str:
        return f"Array(shape={self.shape}, dtype={self.dtype}, name={self.name!r})"


class _BoundedArray(_ArraySpecDM):
    def __init__(self, shape: Any, dtype: Any, minimum: Any, maximum: Any, name: str | None = None):
        super().__init__(shape, dtype, name)
        self.minimum = np.asarray(minimum, dtype=self.dtype)
        self.maximum = np.asarray(maximum, dtype=self.dtype)

    def __repr__(self) -> str:
        return (f"BoundedArray(shape={self.shape}, dtype={self.dtype}, name={self.name!r}, "
                f"minimum={self.minimum}, maximum={self.maximum})")


class _DiscreteArray(_BoundedArray):
    def __init__(self, num_values: int, dtype: Any = np.int32, name: str | None = None):
        super().__init__((), dtype, 0, num_values - 1, name)
        self.num_values = int(num_values)

    def __repr__(self) -> str:
        return f"DiscreteArray(num_values={self.num_values}, dtype={self.dtype}, name={self.name!r})"


class _DMSpecs:
    Array = _ArraySpecDM
    BoundedArray = _BoundedArray
    DiscreteArray = _DiscreteArray


class _DMEnvironment:
    """Stand-in for dm_env.Environment (only used as a base class)."""


if HAVE_DM_ENV:  # pragma: no cover
    TimeStep = dm_env.TimeStep
    StepType = dm_env.StepType
    dm_specs = dm_env.specs
    DMEnvBase = dm_env.Environment
else:
    TimeStep = _TimeStep
    StepType = _StepType
    dm_specs = _DMSpecs
    DMEnvBase = _DMEnvironment
