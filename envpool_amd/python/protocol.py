"""ArraySpec: python view of one exported spec tuple
(mirrors envpool/python/protocol.py:108-136)."""

from __future__ import annotations

from typing import Any

import numpy as np


class ArraySpec:
    """Spec of a numpy array: (dtype, shape, bounds, elementwise bounds, discrete)."""

    def __init__(
        self,
        dtype: type,
        shape: list[int],
        bounds: tuple[Any, Any],
        element_wise_bounds: tuple[Any, Any],
        is_discrete: bool = False,
    ):
        self.dtype = dtype
        self.shape = shape
        self.is_discrete = is_discrete
        if element_wise_bounds[0]:
            self.minimum = np.array(element_wise_bounds[0])
        else:
            self.minimum = bounds[0]
        if element_wise_bounds[1]:
            self.maximum = np.array(element_wise_bounds[1])
        else:
            self.maximum = bounds[1]

    def __repr__(self) -> str:
        return (
            f"ArraySpec(shape={self.shape}, dtype={self.dtype}, "
            f"minimum={self.minimum}, maximum={self.maximum})"
        )
