"""`DevicePool`: thin object over the C ABI for one env family.

It plays the role of the C++ `EnvPool<Spec>` virtual interface of the reference
(envpool/core/envpool.h:29-56: Send / Recv / Reset) — the thing
`PyEnvPool` wraps — with the thread pool replaced by batched HIP kernels.
"""

from __future__ import annotations

import collections
import ctypes
import threading
import weakref
from typing import Any, Sequence

import numpy as np

from . import native


class _PinnedBlocks:
    """Free list of pinned host blocks (epa_host_alloc) that back the arrays
    `recv` returns.  A block is lent to ONE batch: the per-key numpy arrays are
    views of it, and it only comes back here when the last of them is garbage
    collected — so, like the reference's capsule-owned buffers
    (py_envpool.h:40-49), an array is never overwritten by a later step."""

    _MAX_FREE = 4  # per size

    def __init__(self, lib: Any) -> None:
        self._lib = lib
        self._free: dict[int, list[int]] = collections.defaultdict(list)
        self._lock = threading.Lock()
        self._closed = False

    def take(self, nbytes: int) -> np.ndarray:
        with self._lock:
            free = self._free[nbytes]
            ptr = free.pop() if free else None
        if ptr is None:
            ptr = self._lib.epa_host_alloc(nbytes)
            if not ptr:
                raise MemoryError(f"epa_host_alloc({nbytes}) failed")
        base = np.ctypeslib.as_array((ctypes.c_ubyte * nbytes).from_address(ptr))
        fin = weakref.finalize(base, self._give_back, nbytes, ptr)
        fin.atexit = False  # process teardown frees pinned memory anyway
        return base

    def _give_back(self, nbytes: int, ptr: int) -> None:
        with self._lock:
            if not self._closed and len(self._free[nbytes]) < self._MAX_FREE:
                self._free[nbytes].append(ptr)
                return
        self._lib.epa_host_free(ptr)

    def close(self) -> None:
        with self._lock:
            self._closed = True
            ptrs = [p for v in self._free.values() for p in v]
            self._free.clear()
        for p in ptrs:
            self._lib.epa_host_free(p)


class DevicePool:
    """Low-level pool: numpy in, list-of-numpy out, in `_state_keys` order."""

    _SMALL_BATCH_BYTES = 256 * 1024

    def __init__(
        self,
        family: str,
        num_envs: int,
        batch_size: int = 0,
        seed: int = 42,
        env_seed: Sequence[int] | None = None,
        max_episode_steps: int = 0,
        device: int = 0,
        env_id_offset: int = 0,
        params: dict[str, float] | None = None,
    ) -> None:
        self._lib = native.lib()
        self.family = family
        self.num_envs = int(num_envs)
        self.batch_size = int(batch_size) if batch_size else int(num_envs)
        self.env_id_offset = int(env_id_offset)
        self.device = int(device)
        cfg, keep = native.make_config(
            num_envs, batch_size, seed, env_seed, max_episode_steps, device,
            env_id_offset, params,
        )
        h = self._create(family, cfg, params)
        del keep
        self._h = h
        self._pending: collections.deque[int] = collections.deque()
        # per pending send / reset, the pinned block named at send time (None: recv takes one)
        self._posted: collections.deque[Any] = collections.deque()
        self._is_sync = self.batch_size == self.num_envs
        self._blocks = _PinnedBlocks(self._lib)
        self._step_ptrs = None  # step_device: reusable ctypes output array + cache of pointer lists
        self._layouts: dict[int, tuple[list[int], int]] = {}

    def _create(self, family: str, cfg: Any, params: dict[str, float] | None) -> ctypes.c_void_p:
        """epa_create + the key tables (overridden by families with their own constructor)."""
        self.state_keys = native.describe(family, params, "state")
        self.action_keys = native.describe(family, params, "action")
        self.action_dtype = self.action_keys[-1][1]
        self.action_shape = self.action_keys[-1][2]
        h = ctypes.c_void_p()
        native.check(
            self._lib.epa_create(family.encode(), ctypes.byref(cfg), ctypes.byref(h))
        )
        return h

    # -- host path ---------------------------------------------------------
    def send(self, env_id: np.ndarray, action: np.ndarray, post_block: bool = True) -> None:
        env_id = np.ascontiguousarray(env_id, dtype=np.int32)
        action = np.ascontiguousarray(action, dtype=self.action_dtype)
        k = int(env_id.shape[0])
        want = (k, *self.action_shape)
        if action.size != int(np.prod(want)):
            raise RuntimeError(
                f"Expected action of shape {want}, got {action.shape}"
            )
        # A whole-pool step of a sync pool names the block its results shall land in NOW (epa_send_into): the step
        # kernel then writes them straight into it and recv only waits for the kernel.  Whether the engine takes the
        # offer (ids in order, pinned block, "direct_out") is its business: recv hands the same block to
        # epa_recv_block either way.
        block = None
        if post_block and k == (self.num_envs if self._is_sync else self.batch_size):
            _, total = self._layout(k)
            if total >= self._SMALL_BATCH_BYTES:
                block = self._blocks.take(total)
        if block is not None:
            native.check(self._lib.epa_send_into(self._h, env_id.ctypes.data, k, action.ctypes.data,
                                                 block.ctypes.data, block.nbytes))
        else:
            native.check(
                self._lib.epa_send(self._h, env_id.ctypes.data, k, action.ctypes.data)
            )
        if k > 0:  # an empty send enqueues nothing (Pool::Send returns early)
            self._pending.append(k)
            self._posted.append(block)

    def pop_pending(self) -> None:
        """A batch was received through another recv entry point (epa_recv_into of the sharded pool)."""
        self._pending.popleft()
        if self._posted:
            self._posted.popleft()

    def reset(self, env_ids: np.ndarray) -> None:
        env_ids = np.ascontiguousarray(env_ids, dtype=np.int32)
        k = int(env_ids.shape[0])
        native.check(self._lib.epa_reset(self._h, env_ids.ctypes.data, k))
        if k > 0:
            self._pending.append(k)
            self._posted.append(None)

    def _layout(self, rows: int) -> tuple[list[int], int]:
        lay = self._layouts.get(rows)
        if lay is None:
            n = len(self.state_keys)
            offs = (ctypes.c_size_t * n)()
            total = ctypes.c_size_t(0)
            native.check(self._lib.epa_recv_layout(self._h, rows, offs, n, ctypes.byref(total)))
            lay = ([int(o) for o in offs], int(total.value))
            self._layouts[rows] = lay
        return lay

    def recv(self) -> list[np.ndarray]:
        """One device->host copy into a pinned block; the returned arrays are
        views of that block (no host-side memcpy) and own it jointly."""
        if self._is_sync:
            cap = self._pending[0] if self._pending else self.num_envs
        else:
            cap = self.batch_size
        n = len(self.state_keys)
        _, total = self._layout(cap)
        small = total < self._SMALL_BATCH_BYTES
        k = ctypes.c_int32(0)
        if small:
            # tiny batches: fresh pageable arrays + epa_recv's memcpy out of its own
            # pinned landing buffer is cheaper than block bookkeeping
            outs = [np.empty((cap, *shape), dtype=dtype) for _, dtype, shape in self.state_keys]
            ptrs = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs])
            native.check(self._lib.epa_recv(self._h, ptrs, n, cap, ctypes.byref(k)))
        else:
            # the block named at send time for exactly these rows (async: the oldest send is a whole, untouched batch)
            block = None
            if self._posted and (self._is_sync or (self._pending and self._pending[0] == cap)):
                block = self._posted[0]
            if block is None or block.nbytes < total:
                block = self._blocks.take(total)
            offs = (ctypes.c_size_t * n)()
            native.check(
                self._lib.epa_recv_block(self._h, block.ctypes.data, block.nbytes, offs, n,
                                         ctypes.byref(k))
            )
        if self._is_sync:
            if self._pending:
                self._pending.popleft()
            if self._posted:
                self._posted.popleft()
        else:
            # async: rows drain across submissions in order
            left = k.value
            while left > 0 and self._pending:
                if self._pending[0] <= left:
                    left -= self._pending.popleft()
                    if self._posted:
                        self._posted.popleft()
                else:
                    self._pending[0] -= left
                    left = 0
        rows = k.value
        if small:
            return outs if rows == cap else [o[:rows] for o in outs]
        # one view per key straight onto the block (each holds the block as its base: the block goes back to the free
        # list when the last of them dies)
        return [np.ndarray((rows, *shape), dtype=dtype, buffer=block, offset=int(off))
                for (_, dtype, shape), off in zip(self.state_keys, offs)]

    def recv_dict(self) -> dict[str, np.ndarray]:
        return {k[0]: v for k, v in zip(self.state_keys, self.recv())}

    # -- device path ---------------------------------------------------------
    def send_device(self, d_action: int | None, k: int | None = None,
                    d_env_id: int | None = None, wait_event: int | None = None) -> None:
        """`d_action` / `d_env_id` are raw device addresses (ints); `wait_event` is a
        hipEvent_t (int) the producer of those buffers recorded on its stream."""
        k = self.num_envs if k is None else int(k)
        native.check(
            self._lib.epa_send_device(
                self._h, ctypes.c_void_p(d_env_id), k, ctypes.c_void_p(d_action),
                ctypes.c_void_p(wait_event),
            )
        )

    def wait_stream(self, producer_stream: int | None) -> None:
        """Order the pool's stream behind everything enqueued on `producer_stream`
        (raw hipStream_t, e.g. torch.cuda.current_stream().cuda_stream)."""
        native.check(self._lib.epa_wait_stream(self._h, ctypes.c_void_p(producer_stream)))

    def consumer_wait(self, consumer_stream: int | None) -> None:
        """`consumer_stream` waits for the batch the last recv_device handed out."""
        native.check(self._lib.epa_consumer_wait(self._h, ctypes.c_void_p(consumer_stream)))

    def recv_device(self) -> tuple[list[int], int]:
        n = len(self.state_keys)
        ptrs = (ctypes.c_void_p * n)()
        k = ctypes.c_int32(0)
        native.check(self._lib.epa_recv_device(self._h, ptrs, n, ctypes.byref(k)))
        return [int(p) if p else 0 for p in ptrs], k.value

    def step_device(self, d_action: int | None, k: int | None = None, d_env_id: int | None = None,
                    wait_event: int | None = None) -> tuple[tuple[int, ...], int]:
        """send_device + recv_device in one library call (the sync `step()` of the device path).  A pool hands out
        its result blocks in rotation, so the pointer tuples are cached per block (immutable: the same object is
        returned for every step on that block)."""
        k = self.num_envs if k is None else int(k)
        n = len(self.state_keys)
        if self._step_ptrs is None:
            self._step_ptrs, self._step_k, self._step_cache = (ctypes.c_void_p * n)(), ctypes.c_int32(0), {}
        ptrs = self._step_ptrs
        rc = self._lib.epa_step_device(self._h, d_env_id, k, d_action, wait_event, ptrs, n, ctypes.byref(self._step_k))
        if rc:
            native.check(rc)
        key = (ptrs[0], ptrs[n - 1])  # first and last section: the block and its layout
        out = self._step_cache.get(key)
        if out is None:
            out = self._step_cache[key] = tuple(int(p) if p else 0 for p in ptrs)
        return out, self._step_k.value

    @property
    def stream(self) -> int:
        return int(self._lib.epa_stream(self._h) or 0)

    def synchronize(self) -> None:
        native.check(self._lib.epa_synchronize(self._h))

    def set_timing(self, on: bool | int) -> None:
        """False / 0 off; True / 1 an event pair per launch; 2 one pair around the window
        (first launch .. kernel_time_ms()), nothing inserted between the launches."""
        native.check(self._lib.epa_set_timing(self._h, int(on)))

    def kernel_time_ms(self) -> tuple[float, int]:
        ms = ctypes.c_double(0)
        n = ctypes.c_int32(0)
        native.check(
            self._lib.epa_kernel_time_ms(self._h, ctypes.byref(ms), ctypes.byref(n))
        )
        return ms.value, n.value

    # -- test hooks ----------------------------------------------------------
    def state_dim(self) -> int:
        d = ctypes.c_int32(0)
        native.check(self._lib.epa_state_dim(self._h, ctypes.byref(d)))
        return d.value

    def get_state(self, env_ids: Any = None) -> np.ndarray:
        ids = self._ids(env_ids)
        out = np.empty((len(ids), self.state_dim()), dtype=np.float64)
        native.check(
            self._lib.epa_get_state(self._h, ids.ctypes.data, len(ids), out.ctypes.data)
        )
        return out

    def set_state(self, state: np.ndarray, env_ids: Any = None) -> None:
        ids = self._ids(env_ids)
        state = np.ascontiguousarray(state, dtype=np.float64)
        assert state.shape == (len(ids), self.state_dim()), state.shape
        native.check(
            self._lib.epa_set_state(self._h, ids.ctypes.data, len(ids), state.ctypes.data)
        )

    def _ids(self, env_ids: Any) -> np.ndarray:
        if env_ids is None:
            return np.arange(
                self.env_id_offset, self.env_id_offset + self.num_envs, dtype=np.int32
            )
        return np.ascontiguousarray(env_ids, dtype=np.int32)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.epa_destroy(self._h)
            self._h = None
        if getattr(self, "_blocks", None) is not None:
            self._blocks.close()  # blocks still lent out are freed when their arrays die

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:
            pass
