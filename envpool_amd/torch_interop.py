"""Zero-copy views of a pool's device-resident result batch as torch tensors
(the analogue of the reference's XLA path, envpool/core/xla.h, without its
host staging).  torch is only imported here."""

from __future__ import annotations

from typing import Any

import numpy as np

_TYPESTR = {np.dtype(np.int32): "<i4", np.dtype(np.float32): "<f4",
            np.dtype(np.float64): "<f8", np.dtype(np.bool_): "|b1",
            np.dtype(np.uint8): "|u1"}


class _DevArray:
    def __init__(self, ptr: int, shape: tuple, dtype: Any) -> None:
        self.__cuda_array_interface__ = {
            "shape": tuple(int(s) for s in shape),
            "typestr": _TYPESTR[np.dtype(dtype)],
            "data": (int(ptr), False),
            "version": 2,
        }


def recv_device_tensors(pool: Any, device: Any = None) -> dict[str, Any]:
    """pool.recv_device() -> {state key: torch tensor aliasing the batch}.
    Valid until the second next recv_device on the pool; consume it on
    `pool.stream` or after `pool.synchronize()`."""
    import torch

    ptrs, k = pool.recv_device()
    dev = torch.device("cuda", pool.device) if device is None else device
    out = {}
    for (name, dtype, shape), ptr in zip(pool.state_keys, ptrs):
        if k == 0:
            continue
        out[name] = torch.as_tensor(_DevArray(ptr, (k, *shape), dtype), device=dev)
    return out
