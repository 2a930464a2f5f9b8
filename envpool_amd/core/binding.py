"""Native (spec, pool) class pairs for the Python adaptors.

In the reference these classes come out of the pybind11 `REGISTER` macro
(envpool/core/py_envpool.h:303-332): `_XxxEnvSpec` exposing
`_config_keys/_default_config_values/_state_keys/_action_keys/_config_values/
_state_spec/_action_spec`, and `_XxxEnvPool` exposing
`_spec/_send/_recv/_reset/_render/_xla`.  Here the same surface is produced by
`make_native_classes(FamilyDef)` on top of the C ABI (core/native.py), so the
adaptors under envpool_amd/python are oblivious to the swap.

Spec tuples follow SpecTupleHelper (py_envpool.h:103-110):
  (np.dtype, shape list, (lo, hi), (elementwise lo[], hi[]), is_discrete)
with the reference's defaults for unbounded specs, i.e.
std::numeric_limits<T>::min()/max() (envpool/core/spec.h:71-72; note that
::min() of a float type is the smallest positive normal, as in the reference).
"""

from __future__ import annotations

import collections
import ctypes
import threading
from dataclasses import dataclass, field
from typing import Any, Callable, Sequence

import numpy as np

from . import native
from .device_pool import DevicePool

INT_MAX = 2**31 - 1
INT_MIN = -(2**31)
F32_MIN, F32_MAX = float(np.finfo(np.float32).tiny), float(np.finfo(np.float32).max)
F64_MIN, F64_MAX = float(np.finfo(np.float64).tiny), float(np.finfo(np.float64).max)

_DEFAULT_BOUNDS = {
    np.dtype(np.int32): (INT_MIN, INT_MAX),
    np.dtype(np.float32): (F32_MIN, F32_MAX),
    np.dtype(np.float64): (F64_MIN, F64_MAX),
    np.dtype(np.bool_): (False, True),
    np.dtype(np.uint8): (0, 255),
}


def spec(dtype: Any, shape: Sequence[int], bounds: tuple | None = None,
         elementwise: tuple | None = None, is_discrete: bool = False) -> tuple:
    dt = np.dtype(dtype)
    if bounds is None:
        bounds = _DEFAULT_BOUNDS[dt]
    if elementwise is None:
        elementwise = ([], [])
    return (dt, list(shape), tuple(bounds), (list(elementwise[0]), list(elementwise[1])),
            bool(is_discrete))


# common_config / common_action_spec / common_state_spec: env_spec.h:26-43
COMMON_CONFIG: list[tuple[str, Any]] = [
    ("num_envs", 1),
    ("batch_size", 0),
    ("num_threads", 0),
    ("max_num_players", 1),
    ("thread_affinity_offset", -1),
    ("base_path", "envpool"),
    ("seed", 42),
    ("env_seed", []),
    ("gym_reset_return_info", True),
    ("max_episode_steps", INT_MAX),
]
# extensions of this engine, appended AFTER the env-specific keys so that the
# reference's prefix of config keys is unchanged
EXTENSION_CONFIG: list[tuple[str, Any]] = [
    ("device", 0),          # HIP device ordinal, or a list of ordinals to shard over
    ("env_id_offset", 0),   # global id of local env 0 (one shard of a bigger pool)
    # how long recv() waits for rows nobody has sent yet: -1 forever, like the reference's blocking Recv
    # (async_envpool.h:169-181), 0 raise RuntimeError at once, > 0 raise after that many milliseconds
    ("recv_timeout_ms", -1),
]
COMMON_ACTION_SPEC = [
    ("env_id", spec(np.int32, [])),
    ("players.env_id", spec(np.int32, [-1])),
]
COMMON_STATE_SPEC = [
    ("info:env_id", spec(np.int32, [])),
    ("info:players.env_id", spec(np.int32, [-1])),
    ("elapsed_step", spec(np.int32, [])),
    ("done", spec(np.bool_, [])),
    ("reward", spec(np.float32, [-1])),
    ("discount", spec(np.float32, [-1], (0.0, 1.0))),
    ("step_type", spec(np.int32, [])),
    ("trunc", spec(np.bool_, [])),
]


@dataclass
class FamilyDef:
    """Python-side description of `XxxEnvFns` of the reference."""

    name: str                       # class stem, e.g. "CartPole" / "GymHalfCheetah"
    native: str                     # family name understood by the C ABI
    default_config: list[tuple[str, Any]]
    state_spec: Callable[[dict], list[tuple[str, tuple]]]
    action_spec: Callable[[dict], list[tuple[str, tuple]]]
    # config -> numeric params forwarded to the C ABI (epa_config.param_*)
    native_params: Callable[[dict], dict[str, float]] = lambda conf: {}
    # config keys accepted for API compatibility but not supported when changed
    unsupported: dict[str, Any] = field(default_factory=dict)
    # (conf, DevicePool kwargs) -> pool, for families with their own constructor (Atari)
    pool_factory: Callable[[dict, dict], Any] | None = None


class _ShardedPools:
    """num_envs split contiguously over several GPUs of one process:
    shard s owns env ids [offset + s*per, offset + (s+1)*per).  Each shard is a
    DevicePool with its own streams and its own host thread, so uploads, step
    kernels and downloads of all GPUs run concurrently; there is no inter-GPU
    traffic on the data path ("host gather").

    recv: every shard copies its rows device->host DIRECTLY into its row range of
    ONE pinned block (epa_recv_into) whenever the batch is grouped by shard (the
    `step(action)` / `reset()` case, ids ascending) -- no host-side scatter.  Only
    batches whose ids interleave shards fall back to a per-shard recv + scatter."""

    def __init__(self, family: str, devices: Sequence[int], num_envs: int, **kw: Any):
        import concurrent.futures

        if num_envs % len(devices) != 0:
            raise ValueError("num_envs must be divisible by the number of devices")
        if kw.get("batch_size", 0) not in (0, num_envs):
            raise ValueError("async mode (batch_size < num_envs) needs a single device")
        self.per = num_envs // len(devices)
        self.offset = kw.pop("env_id_offset", 0)
        env_seed = kw.pop("env_seed", None)
        kw.pop("batch_size", None)
        self.pools = [
            DevicePool(family, self.per, device=d, env_id_offset=self.offset + s * self.per,
                       env_seed=(env_seed[s * self.per:(s + 1) * self.per] if env_seed else None),
                       **kw)
            for s, d in enumerate(devices)
        ]
        self.state_keys = self.pools[0].state_keys
        self._pending: collections.deque = collections.deque()
        # recv() blocks until a send / reset has been split over the shards (a consumer thread may arrive first)
        self._cv = threading.Condition()
        self._timeout_ms = float((kw.get("params") or {}).get("recv_timeout_ms", -1))
        # one host thread per GPU: the C ABI calls release the GIL (ctypes)
        self._exec = concurrent.futures.ThreadPoolExecutor(len(self.pools))
        self._blocks = self.pools[0]._blocks
        self._lib = self.pools[0]._lib

    def _split(self, ids: np.ndarray) -> tuple[list[Any], bool]:
        """Rows of each shard; `grouped` when every shard's rows are one contiguous run
        (then parts are slices)."""
        shard = (ids - self.offset) // self.per
        if len(ids) and np.all(shard[1:] >= shard[:-1]):
            bounds = np.searchsorted(shard, np.arange(len(self.pools) + 1))
            return [slice(int(bounds[s]), int(bounds[s + 1])) for s in range(len(self.pools))], True
        return [np.flatnonzero(shard == s) for s in range(len(self.pools))], False

    @staticmethod
    def _count(part: Any) -> int:
        return part.stop - part.start if isinstance(part, slice) else len(part)

    def _each(self, fn: Callable[[int, DevicePool, Any], Any], parts: list[Any]) -> None:
        futs = [self._exec.submit(fn, s, p, part)
                for s, (p, part) in enumerate(zip(self.pools, parts)) if self._count(part)]
        for f in futs:
            f.result()

    def send(self, env_id: np.ndarray, action: np.ndarray) -> None:
        env_id = np.ascontiguousarray(env_id, dtype=np.int32)
        action = np.asarray(action)
        parts, grouped = self._split(env_id)
        # (a shard's rows land in ITS row range of one block shared by all shards, epa_recv_into: no block to post)
        self._each(lambda s, p, part: p.send(env_id[part], action[part], post_block=False), parts)
        with self._cv:
            self._pending.append((len(env_id), parts, grouped))
            self._cv.notify_all()

    def reset(self, env_ids: np.ndarray) -> None:
        env_ids = np.ascontiguousarray(env_ids, dtype=np.int32)
        parts, grouped = self._split(env_ids)
        self._each(lambda s, p, part: p.reset(env_ids[part]), parts)
        with self._cv:
            self._pending.append((len(env_ids), parts, grouped))
            self._cv.notify_all()

    def recv(self) -> list[np.ndarray]:
        with self._cv:
            timeout = None if self._timeout_ms < 0 else self._timeout_ms / 1000.0
            if not self._cv.wait_for(lambda: bool(self._pending), timeout):
                raise RuntimeError(f"recv: nothing pending (recv_timeout_ms = {self._timeout_ms:g})")
            k, parts, grouped = self._pending.popleft()
        if not grouped:
            outs = [np.empty((k, *shape), dtype=dtype) for _, dtype, shape in self.state_keys]

            def scatter(s: int, p: DevicePool, idx: Any) -> None:
                for o, part in zip(outs, p.recv()):
                    o[idx] = part

            self._each(scatter, parts)
            return outs
        # one pinned block for the whole batch, laid out like a DevicePool batch of k rows
        row_bytes = [int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
                     for _, dtype, shape in self.state_keys]
        offs, total = [], 0
        for rb in row_bytes:
            offs.append(total)
            total += (k * rb + 255) // 256 * 256
        block = self._blocks.take(max(total, 256))
        base = block.ctypes.data
        n = len(self.state_keys)

        def land(s: int, p: DevicePool, part: slice) -> None:
            ptrs = (ctypes.c_void_p * n)(*[base + o + part.start * rb
                                           for o, rb in zip(offs, row_bytes)])
            got = ctypes.c_int32(0)
            native.check(self._lib.epa_recv_into(p._h, ptrs, n, part.stop - part.start,
                                                 ctypes.byref(got)))
            assert got.value == part.stop - part.start
            p.pop_pending()

        self._each(land, parts)
        return [block[o:o + k * rb].view(dtype).reshape((k, *shape))
                for (_, dtype, shape), o, rb in zip(self.state_keys, offs, row_bytes)]

    def close(self) -> None:
        self._exec.shutdown(wait=True)
        for p in self.pools:
            p.close()


def make_native_classes(fd: FamilyDef, static_action_spec: list | None = None) -> tuple[type, type]:
    """Build `_<Name>EnvSpec` and `_<Name>EnvPool` for a family.  `static_action_spec`: key
    list to use for the class-level `_action_keys` when `fd.action_spec` cannot be evaluated
    on the default config (Atari sizes its action set from the ROM)."""
    config_items = COMMON_CONFIG + list(fd.default_config) + EXTENSION_CONFIG
    config_keys = [k for k, _ in config_items]
    default_values = tuple(v for _, v in config_items)
    # key lists are static per class in the reference (py_envpool.h:163-171)
    default_conf = dict(config_items)
    state_keys = [k for k, _ in COMMON_STATE_SPEC] + [k for k, _ in fd.state_spec(default_conf)]
    action_keys = [k for k, _ in COMMON_ACTION_SPEC] + [
        k for k, _ in (static_action_spec or fd.action_spec(default_conf))]

    class _Spec:
        _config_keys = config_keys
        _default_config_values = default_values
        _state_keys = state_keys
        _action_keys = action_keys

        def __init__(self, config_values: Any) -> None:
            values = list(config_values)
            if len(values) != len(config_keys):
                raise TypeError(
                    f"{type(self).__name__} expects {len(config_keys)} config values"
                )
            conf = dict(zip(config_keys, values))
            # EnvSpec ctor, envpool/core/env_spec.h:75-83
            if conf["batch_size"] > conf["num_envs"]:
                raise ValueError(
                    "It is required that batch_size <= num_envs, got num_envs = "
                    f"{conf['num_envs']}, batch_size = {conf['batch_size']}"
                )
            if conf["batch_size"] == 0:
                conf["batch_size"] = conf["num_envs"]
            for key, supported in fd.unsupported.items():
                if conf.get(key, supported) != supported:
                    raise ValueError(
                        f"{fd.name}: {key}={conf[key]!r} is not supported by the "
                        f"MI355X engine yet (only {supported!r})"
                    )
            self._conf = conf
            self._config_values = tuple(conf[k] for k in config_keys)
            # like EnvSpec's ctor (env_spec.h:70-74) the specs are built -- and
            # their arguments validated -- at construction time
            fd.state_spec(conf)
            fd.action_spec(conf)
            fd.native_params(conf)

        @property
        def _state_spec(self) -> tuple:
            return tuple(s for _, s in COMMON_STATE_SPEC) + tuple(
                s for _, s in fd.state_spec(self._conf))

        @property
        def _action_spec(self) -> tuple:
            return tuple(s for _, s in COMMON_ACTION_SPEC) + tuple(
                s for _, s in fd.action_spec(self._conf))

    class _Pool:
        _state_keys = state_keys
        _action_keys = action_keys

        def __init__(self, spec: Any) -> None:
            conf = dict(zip(spec._config_keys, spec._config_values))
            if conf["max_num_players"] != 1:
                raise ValueError("only single-player envs are on the MI355X path")
            params = {k: float(v) for k, v in fd.native_params(conf).items()}
            if conf["recv_timeout_ms"] != -1:
                params["recv_timeout_ms"] = float(conf["recv_timeout_ms"])
            kw = dict(
                batch_size=conf["batch_size"],
                seed=conf["seed"],
                env_seed=list(conf["env_seed"]) or None,
                max_episode_steps=conf["max_episode_steps"],
                env_id_offset=conf["env_id_offset"],
                params=params,
            )
            device = conf["device"]
            if fd.pool_factory is not None:
                if isinstance(device, (list, tuple)):
                    if len(device) != 1:
                        raise ValueError(f"{fd.name}: in-process sharding over several devices "
                                         "is not available for this family")
                    device = device[0]
                self._pool = fd.pool_factory(conf, dict(kw, device=int(device)))
            elif isinstance(device, (list, tuple)) and len(device) > 1:
                self._pool: Any = _ShardedPools(fd.native, list(device), conf["num_envs"], **kw)
            else:
                if isinstance(device, (list, tuple)):
                    device = device[0]
                self._pool = DevicePool(fd.native, conf["num_envs"], device=int(device), **kw)
            self._spec = spec
            # the C ABI's view of the layout must agree with the Python spec
            native_keys = [k for k, _, _ in self._pool.state_keys]
            assert native_keys == state_keys, (native_keys, state_keys)

        def _send(self, action: list[np.ndarray]) -> None:
            # list order = _action_keys: env_id, players.env_id, <env action>
            self._pool.send(action[0], action[-1])

        def _recv(self) -> list[np.ndarray]:
            return self._pool.recv()

        def _reset(self, env_ids: np.ndarray) -> None:
            self._pool.reset(np.asarray(env_ids, dtype=np.int32))

        def _render(self, env_ids: np.ndarray, width: int, height: int,
                    camera_id: int) -> np.ndarray:
            # async_envpool.h:192-194
            raise RuntimeError("render not implemented for this environment")

        def _xla(self) -> Any:
            raise RuntimeError("XLA is not available for the MI355X engine")

        def close(self) -> None:
            self._pool.close()

        @property
        def device_pool(self) -> Any:
            """Extension: the underlying DevicePool (zero-copy device path)."""
            return self._pool

    _Spec.__name__ = _Spec.__qualname__ = f"_{fd.name}EnvSpec"
    _Pool.__name__ = _Pool.__qualname__ = f"_{fd.name}EnvPool"
    return _Spec, _Pool
