"""The real-ALE emulator plugin (integration/ale_adapter/ale_adapter.cc) is compiled code: built
here against the ALE-API shim the repository owns (oracle/ref_shims_atari/ale_interface.hpp, the
members the reference's atari_env.h touches, over the synthetic console) and driven through the
plugin ABI (include/envpool_amd_emulator.h) next to the synthetic plugin -- same console behind two
different API surfaces, so every call must agree byte for byte.  CPU only; the `-m gpu` leg
(tests/test_gpu_atari_env.py) runs the reference-generated fixtures through both plugins."""
import ctypes
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from atari_util import ROMS, adapter_path, plugin_path  # noqa: E402


class _Cfg(ctypes.Structure):
    _fields_ = [("rom_path", ctypes.c_char_p), ("random_seed", ctypes.c_int32),
                ("repeat_action_probability", ctypes.c_float), ("mode", ctypes.c_int32),
                ("difficulty", ctypes.c_int32)]


_P = ctypes.c_void_p
_U8P = ctypes.POINTER(ctypes.c_uint8)


class _Api(ctypes.Structure):
    _fields_ = [
        ("abi", ctypes.c_int32),
        ("create", ctypes.CFUNCTYPE(_P, ctypes.POINTER(_Cfg))),
        ("destroy", ctypes.CFUNCTYPE(None, _P)),
        ("action_set", ctypes.CFUNCTYPE(ctypes.c_int32, _P, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32)),
        ("reset_game", ctypes.CFUNCTYPE(None, _P)),
        ("act", ctypes.CFUNCTYPE(ctypes.c_int32, _P, ctypes.c_int32)),
        ("game_over", ctypes.CFUNCTYPE(ctypes.c_int32, _P)),
        ("lives", ctypes.CFUNCTYPE(ctypes.c_int32, _P)),
        ("screen", ctypes.CFUNCTYPE(_U8P, _P)),
        ("ram", ctypes.CFUNCTYPE(_U8P, _P)),
        ("palette", ctypes.CFUNCTYPE(None, _P, _U8P, _U8P)),
        ("last_error", ctypes.CFUNCTYPE(ctypes.c_char_p)),
    ]


def _api(path):
    lib = ctypes.CDLL(path)
    lib.epa_emulator_get_api.restype = ctypes.POINTER(_Api)
    return lib, lib.epa_emulator_get_api().contents


def test_adapter_exports_the_plugin_abi():
    _, api = _api(adapter_path())
    assert api.abi == 1  # EPA_EMULATOR_ABI
    bad = _Cfg(b"/synthetic/atari/roms/no_such_rom.bin", 0, 0.0, -1, -1)
    assert not api.create(ctypes.byref(bad))  # loadROM throws -> NULL + message
    assert b"ROM" in api.last_error()


@pytest.mark.parametrize("rom", ROMS)
@pytest.mark.parametrize("sticky,mode", [(0.0, -1), (0.25, 1)])
def test_adapter_equals_synthetic_plugin_call_by_call(rom, sticky, mode):
    keep = []
    pair = []
    for path in (plugin_path(), adapter_path()):
        lib, api = _api(path)
        keep.append(lib)
        cfg = _Cfg(f"/synthetic/atari/roms/{rom}.bin".encode(), 123, sticky, mode, 0 if mode >= 0 else -1)
        h = api.create(ctypes.byref(cfg))
        assert h, api.last_error()
        pair.append((api, h))

    def both(f):
        a, b = (f(api, h) for api, h in pair)
        return a, b

    def arr(ptr, n):
        return np.ctypeslib.as_array(ptr, shape=(n,)).copy()

    for full in (0, 1):
        def aset(api, h):
            codes = (ctypes.c_int32 * 32)()
            k = api.action_set(h, full, codes, 32)
            return list(codes[:k])
        a, b = both(aset)
        assert a == b and len(a) > 0

    def pal(api, h):
        g = (ctypes.c_uint8 * 256)()
        rgb = (ctypes.c_uint8 * 768)()
        api.palette(h, g, rgb)
        return bytes(g) + bytes(rgb)
    a, b = both(pal)
    assert a == b
    codes = aset(*pair[0])
    rng = np.random.default_rng(0)
    both(lambda api, h: api.reset_game(h))
    n_over = 0
    for t in range(600):
        act = int(codes[rng.integers(len(codes))])
        ra, rb = both(lambda api, h: api.act(h, act))
        assert ra == rb, t
        sa, sb = both(lambda api, h: arr(api.screen(h), 210 * 160))
        assert np.array_equal(sa, sb), t
        ma, mb = both(lambda api, h: arr(api.ram(h), 128))
        assert np.array_equal(ma, mb), t
        la, lb = both(lambda api, h: (api.lives(h), api.game_over(h)))
        assert la == lb, t
        if la[1]:
            n_over += 1
            both(lambda api, h: api.reset_game(h))
    assert n_over > 0 or rom == "synth_nofire"
    for api, h in pair:
        api.destroy(h)
