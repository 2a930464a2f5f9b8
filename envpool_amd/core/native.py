"""ctypes binding of the C ABI declared in include/envpool_amd.h.

This is the reference-side stub a maintainer would write instead of
`PyEnvPool<AsyncEnvPool<Env>>` (envpool/core/py_envpool.h:206-288): the Python
adaptors keep calling `_send/_recv/_reset`, which land here.

There is NO CPU fallback: if the HIP library is missing, or no GPU is visible
when a pool is created, an exception is raised.
"""

from __future__ import annotations

import ctypes
import os
from typing import Sequence

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# ENVPOOL_AMD_LIB: another build of the same C ABI (the cross-check tests load lib/libenvpool_amd_alt.so, the
# product library plus the superseded Humanoid kernels: `make -C envpool_amd/csrc EPA_ALT_KERNELS=1`)
LIB_PATH = os.environ.get("ENVPOOL_AMD_LIB") or os.path.join(_PKG, "lib", "libenvpool_amd.so")

EPA_OK, EPA_ERR_INVALID, EPA_ERR_RUNTIME, EPA_ERR_DEVICE = 0, 1, 2, 3
DTYPES = {0: np.int32, 1: np.float32, 2: np.float64, 3: np.bool_, 4: np.uint8}


class EpaConfig(ctypes.Structure):
    _fields_ = [
        ("num_envs", ctypes.c_int32),
        ("batch_size", ctypes.c_int32),
        ("seed", ctypes.c_int32),
        ("env_seed", ctypes.POINTER(ctypes.c_int32)),
        ("max_episode_steps", ctypes.c_int32),
        ("device", ctypes.c_int32),
        ("env_id_offset", ctypes.c_int32),
        ("n_params", ctypes.c_int32),
        ("param_keys", ctypes.POINTER(ctypes.c_char_p)),
        ("param_values", ctypes.POINTER(ctypes.c_double)),
    ]


class EpaAtariConfig(ctypes.Structure):
    _fields_ = [
        ("base", EpaConfig),
        ("rom_path", ctypes.c_char_p),
        ("emulator_lib", ctypes.c_char_p),
    ]


class EpaKeyInfo(ctypes.Structure):
    _fields_ = [
        ("name", ctypes.c_char_p),
        ("dtype", ctypes.c_int32),
        ("ndim", ctypes.c_int32),
        ("shape", ctypes.c_int32 * 4),
        ("row_elems", ctypes.c_int32),
        ("row_bytes", ctypes.c_int32),
    ]


_lib: ctypes.CDLL | None = None


def lib() -> ctypes.CDLL:
    """Load libenvpool_amd.so (built by `__graft_entry__.build()`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"envpool_amd: HIP library not built ({LIB_PATH} missing). Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C envpool_amd/csrc`. There is no CPU fallback."
        )
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, cp = ctypes.c_void_p, ctypes.c_int32, ctypes.c_char_p
    P = ctypes.POINTER
    sig = {
        "epa_num_families": (i32, []),
        "epa_family_name": (cp, [i32]),
        "epa_describe_state": (i32, [cp, P(EpaConfig), P(EpaKeyInfo), i32, P(i32)]),
        "epa_describe_action": (i32, [cp, P(EpaConfig), P(EpaKeyInfo), i32, P(i32)]),
        "epa_create": (i32, [cp, P(EpaConfig), P(vp)]),
        "epa_destroy": (i32, [vp]),
        "epa_send": (i32, [vp, vp, i32, vp]),
        "epa_send_into": (i32, [vp, vp, i32, vp, vp, ctypes.c_size_t]),
        "epa_reset": (i32, [vp, vp, i32]),
        "epa_recv": (i32, [vp, P(vp), i32, i32, P(i32)]),
        "epa_recv_layout": (i32, [vp, i32, P(ctypes.c_size_t), i32, P(ctypes.c_size_t)]),
        "epa_recv_block": (i32, [vp, vp, ctypes.c_size_t, P(ctypes.c_size_t), i32, P(i32)]),
        "epa_recv_into": (i32, [vp, P(vp), i32, i32, P(i32)]),
        "epa_pending_rows": (i32, [vp, P(i32)]),
        "epa_send_device": (i32, [vp, vp, i32, vp, vp]),
        "epa_wait_stream": (i32, [vp, vp]),
        "epa_consumer_wait": (i32, [vp, vp]),
        "epa_recv_device": (i32, [vp, P(vp), i32, P(i32)]),
        "epa_step_device": (i32, [vp, vp, i32, vp, vp, P(vp), i32, P(i32)]),
        "epa_stream": (vp, [vp]),
        "epa_synchronize": (i32, [vp]),
        "epa_set_timing": (i32, [vp, i32]),
        "epa_kernel_time_ms": (i32, [vp, P(ctypes.c_double), P(i32)]),
        "epa_state_dim": (i32, [vp, P(i32)]),
        "epa_get_state": (i32, [vp, vp, i32, vp]),
        "epa_set_state": (i32, [vp, vp, i32, vp]),
        "epa_atari_post_create": (i32, [i32] * 8 + [P(vp)]),
        "epa_atari_post_create_ex": (i32, [i32] * 8 + [vp, i32, P(vp)]),
        "epa_atari_create": (i32, [P(EpaAtariConfig), P(vp)]),
        "epa_atari_num_actions": (i32, [P(EpaAtariConfig), P(i32)]),
        "epa_pool_state_keys": (i32, [vp, P(EpaKeyInfo), i32, P(i32)]),
        "epa_pool_action_keys": (i32, [vp, P(EpaKeyInfo), i32, P(i32)]),
        "epa_atari_post_destroy": (i32, [vp]),
        "epa_atari_post_push": (i32, [vp, vp, i32, vp, vp, vp]),
        "epa_atari_post_push_device": (i32, [vp, vp, i32, vp, vp, vp]),
        "epa_atari_post_stream": (vp, [vp]),
        "epa_last_error": (cp, []),
        "epa_version": (cp, []),
        "epa_device_count": (i32, [P(i32)]),
        "epa_host_alloc": (vp, [ctypes.c_size_t]),
        "epa_host_free": (None, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


EXPORTED_SYMBOLS = [
    "epa_num_families", "epa_family_name", "epa_describe_state",
    "epa_describe_action", "epa_create", "epa_destroy", "epa_send", "epa_reset",
    "epa_recv", "epa_recv_layout", "epa_recv_block", "epa_send_into", "epa_recv_into", "epa_pending_rows",
    "epa_send_device", "epa_recv_device", "epa_step_device", "epa_wait_stream", "epa_consumer_wait",
    "epa_stream", "epa_synchronize", "epa_set_timing", "epa_kernel_time_ms",
    "epa_state_dim", "epa_get_state", "epa_set_state", "epa_atari_post_create",
    "epa_atari_post_create_ex", "epa_atari_create", "epa_atari_num_actions",
    "epa_pool_state_keys", "epa_pool_action_keys",
    "epa_atari_post_destroy", "epa_atari_post_push",
    "epa_atari_post_push_device", "epa_atari_post_stream", "epa_last_error",
    "epa_version", "epa_device_count", "epa_host_alloc", "epa_host_free",
]


def check(code: int) -> None:
    """Map C-ABI error classes onto the exceptions the reference raises
    (std::invalid_argument -> ValueError, std::runtime_error -> RuntimeError)."""
    if code == EPA_OK:
        return
    msg = lib().epa_last_error().decode()
    if code == EPA_ERR_INVALID:
        raise ValueError(msg)
    raise RuntimeError(msg)


def device_count() -> int:
    n = ctypes.c_int32(0)
    check(lib().epa_device_count(ctypes.byref(n)))
    return n.value


def make_config(
    num_envs: int,
    batch_size: int = 0,
    seed: int = 42,
    env_seed: Sequence[int] | None = None,
    max_episode_steps: int = 0,
    device: int = 0,
    env_id_offset: int = 0,
    params: dict[str, float] | None = None,
) -> tuple[EpaConfig, list]:
    """Build an epa_config; returns (config, keepalive objects)."""
    keep: list = []
    cfg = EpaConfig()
    cfg.num_envs = num_envs
    cfg.batch_size = batch_size
    cfg.seed = seed
    if env_seed is not None and len(env_seed) > 0:
        arr = (ctypes.c_int32 * len(env_seed))(*[int(s) for s in env_seed])
        keep.append(arr)
        cfg.env_seed = ctypes.cast(arr, ctypes.POINTER(ctypes.c_int32))
    cfg.max_episode_steps = int(min(max_episode_steps, 2**31 - 1))
    cfg.device = device
    cfg.env_id_offset = env_id_offset
    params = params or {}
    cfg.n_params = len(params)
    if params:
        keys = (ctypes.c_char_p * len(params))(*[k.encode() for k in params])
        vals = (ctypes.c_double * len(params))(*[float(v) for v in params.values()])
        keep += [keys, vals]
        cfg.param_keys = ctypes.cast(keys, ctypes.POINTER(ctypes.c_char_p))
        cfg.param_values = ctypes.cast(vals, ctypes.POINTER(ctypes.c_double))
    return cfg, keep


def pool_keys(handle: ctypes.c_void_p, which: str = "state"):
    """[(name, np dtype, row shape tuple)] of an existing pool's state or action keys."""
    keys = (EpaKeyInfo * 32)()
    n = ctypes.c_int32(0)
    fn = lib().epa_pool_state_keys if which == "state" else lib().epa_pool_action_keys
    check(fn(handle, keys, 32, ctypes.byref(n)))
    return [(keys[i].name.decode(), DTYPES[keys[i].dtype], tuple(keys[i].shape[: keys[i].ndim]))
            for i in range(n.value)]


def describe(family: str, params: dict[str, float] | None = None, which: str = "state"):
    """[(name, np dtype, row shape tuple)] of a family's state or action keys."""
    cfg, keep = make_config(1, params=params)
    keys = (EpaKeyInfo * 32)()
    n = ctypes.c_int32(0)
    fn = lib().epa_describe_state if which == "state" else lib().epa_describe_action
    check(fn(family.encode(), ctypes.byref(cfg), keys, 32, ctypes.byref(n)))
    out = []
    for i in range(n.value):
        k = keys[i]
        out.append((k.name.decode(), DTYPES[k.dtype], tuple(k.shape[: k.ndim])))
    del keep
    return out
