"""Zero-copy views of a pool's device-resident result batch as torch tensors
(the analogue of the reference's XLA path, envpool/core/xla.h, without its
host staging).  torch is only imported here."""

from __future__ import annotations

import collections
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


def send_device_tensors(pool: Any, action: Any, env_id: Any = None) -> None:
    """Step the envs with an action tensor that lives on the pool's device.  The step
    kernel is ordered behind everything enqueued so far on torch's CURRENT stream
    (the stream the learner produced `action` on) -- no host synchronisation.
    `action=None` resets the listed envs."""
    import torch

    k = None
    d_action = None
    if action is not None:
        if not action.is_contiguous() or action.dtype != _torch_dtype(pool.action_dtype):
            raise RuntimeError(
                f"send_device_tensors: action must be contiguous {np.dtype(pool.action_dtype)}")
        d_action = action.data_ptr()
        k = int(action.shape[0])
    d_ids = None
    if env_id is not None:
        if env_id.dtype != torch.int32 or not env_id.is_contiguous():
            raise RuntimeError("send_device_tensors: env_id must be contiguous int32")
        d_ids = env_id.data_ptr()
        k = int(env_id.shape[0])
    dev = torch.device("cuda", pool.device)
    pool.wait_stream(torch.cuda.current_stream(dev).cuda_stream)
    pool.send_device(d_action, k, d_ids)
    # The step kernel reads both tensors on the pool's PRIVATE stream, which torch's caching
    # allocator knows nothing about: a temporary the caller drops right after this call
    # (`send_device_tensors(pool, policy(obs))`) could be handed to a later kernel on torch's stream
    # while the step kernel is still reading it.  So the pool keeps a reference to what it was
    # sent until an event recorded behind the step kernel has passed.  (Not `Tensor.record_stream`:
    # that makes the allocator record an event on the pool's stream when the tensor is freed --
    # possibly after the pool, and its stream, are gone.)
    sent = pool.__dict__.setdefault("_sent_tensors", collections.deque())
    while sent and sent[0][0].query():
        sent.popleft()
    if action is not None or env_id is not None:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.ExternalStream(pool.stream, device=dev))
        sent.append((ev, action, env_id))


def _torch_dtype(dt: Any) -> Any:
    import torch

    return {np.dtype(np.int32): torch.int32, np.dtype(np.float32): torch.float32,
            np.dtype(np.float64): torch.float64}[np.dtype(dt)]


def recv_device_tensors(pool: Any, device: Any = None, order_current_stream: bool = True
                        ) -> dict[str, Any]:
    """pool.recv_device() -> {state key: torch tensor aliasing the batch}.
    Valid until the second next recv_device on the pool.  With
    `order_current_stream` torch's current stream is made to wait for the step
    kernel, so the tensors can be consumed right away without a host sync."""
    import torch

    ptrs, k = pool.recv_device()
    dev = torch.device("cuda", pool.device) if device is None else device
    if order_current_stream:
        pool.consumer_wait(torch.cuda.current_stream(dev).cuda_stream)
    out = {}
    for (name, dtype, shape), ptr in zip(pool.state_keys, ptrs):
        if k == 0:
            continue
        out[name] = torch.as_tensor(_DevArray(ptr, (k, *shape), dtype), device=dev)
    return out
