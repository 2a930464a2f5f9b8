"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

ctypes wrapper over the small `orc_*` C API that both oracles export:

* ``oracle/_ref/libref_oracle.so``  — the reference's own C++ compiled in place
  from /root/reference (kind "reference"; exists only where it was built).
* ``oracle/_ref/libref_atari.so``   — the reference's own atari_env.h compiled in place
  over the synthetic console of tests/synth_ale (kind "reference_atari").
* ``oracle/_ref/libref_mujoco.so``  — the reference's own gym-MuJoCo task wrappers
  (envpool/mujoco/gym/*.h) inside its own AsyncEnvPool, compiled in place over a mujoco.h
  shim whose engine calls forward to oracle/mjcpu (kind "reference_mujoco": reference
  wrapper + runtime, ported engine).
* ``oracle/_build/liboracle.so``    — the plain-C restatement under
  ``oracle/restate`` and ``oracle/mjcpu`` (kind "port"; travels everywhere).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline
leg may import this module; nothing under ``envpool_amd/`` does.
"""

from __future__ import annotations

import ctypes
import os
from typing import Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_LIB = os.path.join(_HERE, "_ref", "libref_oracle.so")
REF_ATARI_LIB = os.path.join(_HERE, "_ref", "libref_atari.so")
REF_MUJOCO_LIB = os.path.join(_HERE, "_ref", "libref_mujoco.so")
PORT_LIB = os.path.join(_HERE, "_build", "liboracle.so")

_DTYPES = {0: np.int32, 1: np.float32, 2: np.float64, 3: np.bool_, 4: np.uint8}

_libs: dict[str, ctypes.CDLL] = {}


def _load(path: str) -> ctypes.CDLL:
    if path not in _libs:
        lib = ctypes.CDLL(path)
        lib.orc_create.restype = ctypes.c_void_p
        lib.orc_create.argtypes = [
            ctypes.c_char_p,
            ctypes.c_int,
            ctypes.c_int,
            ctypes.c_int,
            ctypes.POINTER(ctypes.c_double),
            ctypes.c_int,
            ctypes.c_int,
        ]
        lib.orc_num_state_keys.argtypes = [ctypes.c_void_p]
        lib.orc_state_key.argtypes = [
            ctypes.c_void_p,
            ctypes.c_int,
            ctypes.c_char_p,
            ctypes.POINTER(ctypes.c_int),
            ctypes.POINTER(ctypes.c_int),
        ]
        lib.orc_action_info.argtypes = [
            ctypes.c_void_p,
            ctypes.POINTER(ctypes.c_int),
            ctypes.POINTER(ctypes.c_int),
        ]
        lib.orc_reset.argtypes = [
            ctypes.c_void_p,
            ctypes.c_void_p,
            ctypes.c_int,
            ctypes.POINTER(ctypes.c_void_p),
        ]
        lib.orc_reset.restype = None
        lib.orc_step.argtypes = [
            ctypes.c_void_p,
            ctypes.c_void_p,
            ctypes.c_int,
            ctypes.c_void_p,
            ctypes.POINTER(ctypes.c_void_p),
        ]
        lib.orc_step.restype = None
        lib.orc_time_steps.argtypes = [
            ctypes.c_void_p,
            ctypes.c_int,
            ctypes.c_void_p,
        ]
        lib.orc_time_steps.restype = ctypes.c_double
        lib.orc_destroy.argtypes = [ctypes.c_void_p]
        lib.orc_destroy.restype = None
        lib.orc_kind.restype = ctypes.c_char_p
        _libs[path] = lib
    return _libs[path]


def have_ref() -> bool:
    return os.path.exists(REF_LIB)


def have_ref_atari() -> bool:
    return os.path.exists(REF_ATARI_LIB)


def have_ref_mujoco() -> bool:
    return os.path.exists(REF_MUJOCO_LIB)


def have_port() -> bool:
    return os.path.exists(PORT_LIB)


class Oracle:
    """One oracle pool. ``kind``: "reference", "reference_atari", "reference_mujoco", "port"."""

    def __init__(
        self,
        task: str,
        num_envs: int,
        seed: int = 0,
        max_episode_steps: int = 0,
        extra: Sequence[float] = (),
        kind: str = "port",
        num_threads: int = 1,
    ) -> None:
        path = {
            "reference": REF_LIB,
            "reference_atari": REF_ATARI_LIB,
            "reference_mujoco": REF_MUJOCO_LIB,
            "port": PORT_LIB,
        }[kind]
        self.lib = _load(path)
        self.kind = self.lib.orc_kind().decode()
        ex = (ctypes.c_double * max(1, len(extra)))(*extra)
        self.h = self.lib.orc_create(
            task.encode(),
            num_envs,
            seed,
            max_episode_steps,
            ex,
            len(extra),
            num_threads,
        )
        if not self.h:
            raise RuntimeError(f"oracle {kind}: cannot create {task}")
        self.num_envs = num_envs
        self.keys: list[tuple[str, type, int]] = []
        name = ctypes.create_string_buffer(64)
        dt = ctypes.c_int()
        el = ctypes.c_int()
        for i in range(self.lib.orc_num_state_keys(self.h)):
            self.lib.orc_state_key(self.h, i, name, dt, el)
            self.keys.append((name.value.decode(), _DTYPES[dt.value], el.value))
        self.lib.orc_action_info(self.h, dt, el)
        self.action_dtype = _DTYPES[dt.value]
        self.action_elems = el.value

    def _alloc(self, k: int) -> tuple[dict[str, np.ndarray], ctypes.Array]:
        out = {}
        ptrs = (ctypes.c_void_p * len(self.keys))()
        for i, (name, dtype, elems) in enumerate(self.keys):
            # zero-initialised like the reference's fresh StateBuffer
            arr = np.zeros((k, elems), dtype=dtype)
            out[name] = arr
            ptrs[i] = arr.ctypes.data
        return out, ptrs

    def reset(self, ids: np.ndarray | None = None) -> dict[str, np.ndarray]:
        if ids is None:
            ids = np.arange(self.num_envs, dtype=np.int32)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        out, ptrs = self._alloc(len(ids))
        self.lib.orc_reset(self.h, ids.ctypes.data, len(ids), ptrs)
        return out

    def step(
        self, action: np.ndarray, ids: np.ndarray | None = None
    ) -> dict[str, np.ndarray]:
        if ids is None:
            ids = np.arange(self.num_envs, dtype=np.int32)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        action = np.ascontiguousarray(action, dtype=self.action_dtype)
        assert action.size == len(ids) * self.action_elems, action.shape
        out, ptrs = self._alloc(len(ids))
        self.lib.orc_step(
            self.h, ids.ctypes.data, len(ids), action.ctypes.data, ptrs
        )
        return out

    def time_steps(self, steps: int, action: np.ndarray) -> float:
        action = np.ascontiguousarray(action, dtype=self.action_dtype)
        assert action.size == self.num_envs * self.action_elems
        return self.lib.orc_time_steps(self.h, steps, action.ctypes.data)

    # teacher-forcing hooks (port oracle only)
    def state_dim(self) -> int:
        self.lib.orc_state_dim.argtypes = [ctypes.c_void_p]
        return self.lib.orc_state_dim(self.h)

    def get_state(self, ids: np.ndarray | None = None) -> np.ndarray:
        if ids is None:
            ids = np.arange(self.num_envs, dtype=np.int32)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        out = np.zeros((len(ids), self.state_dim()), dtype=np.float64)
        self.lib.orc_get_state.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int, ctypes.c_void_p]
        self.lib.orc_get_state.restype = None
        self.lib.orc_get_state(self.h, ids.ctypes.data, len(ids), out.ctypes.data)
        return out

    def set_state(self, state: np.ndarray, ids: np.ndarray | None = None) -> None:
        if ids is None:
            ids = np.arange(self.num_envs, dtype=np.int32)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        state = np.ascontiguousarray(state, dtype=np.float64)
        self.lib.orc_set_state.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int, ctypes.c_void_p]
        self.lib.orc_set_state.restype = None
        self.lib.orc_set_state(self.h, ids.ctypes.data, len(ids), state.ctypes.data)

    def close(self) -> None:
        if self.h:
            self.lib.orc_destroy(self.h)
            self.h = None

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:
            pass
