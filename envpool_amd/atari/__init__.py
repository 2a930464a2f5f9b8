"""Atari (mirror of envpool/atari/__init__.py).

The reference's AtariEnv (envpool/atari/atari_env.h) runs ALE on the host and
then max-pools the last two frames, resizes to 84x84 (cv::resize INTER_AREA)
and pushes into a 4-deep frame stack (`PushStack`, atari_env.h:308-346).  Here:

* `AtariEnvSpec / AtariDMEnvPool / AtariGymnasiumEnvPool`: the full env.  The emulator
  stays on the host (north star) behind a plugin (`emulator_lib`, see
  include/envpool_amd_emulator.h): host worker threads run the frame_skip loop, the colour
  palette + max-pool + resize + transpose + frame stack run as one HIP kernel per batch.
  ALE and its ROMs are not part of this repository: the plugin for the real ALE is
  integration/ale_adapter (built by the deployer), tests use tests/synth_ale.
* `AtariPostProcess`: the stand-alone post-process (K4) for callers with their own
  emulator loop.  gray_scale / RGB, INTER_AREA (the default `use_inter_area_resize=True`)
  or INTER_LINEAR (`False`, the reference benchmark's setting).
"""

from __future__ import annotations

import ctypes
import os
from typing import Any

import numpy as np

from envpool_amd.core import native
from envpool_amd.core.binding import FamilyDef, make_native_classes, spec
from envpool_amd.core.device_pool import DevicePool
from envpool_amd.python.api import py_env

# AtariEnvFns::DefaultConfig, atari_env.h:52-63
_ATARI_CONFIG: list[tuple[str, Any]] = [
    ("stack_num", 4), ("frame_skip", 4), ("noop_max", 30),
    ("zero_discount_on_life_loss", False), ("episodic_life", False),
    ("reward_clip", False), ("use_fire_reset", True),
    ("img_height", 84), ("img_width", 84), ("task", "pong"), ("mode", -1),
    ("difficulty", -1), ("full_action_space", False),
    ("repeat_action_probability", 0.0), ("use_inter_area_resize", True), ("gray_scale", True),
    # extension: the emulator plugin (default: $EPA_ATARI_EMULATOR_LIB)
    ("emulator_lib", ""),
]
_NUMERIC = ("stack_num", "frame_skip", "noop_max", "zero_discount_on_life_loss", "episodic_life",
            "reward_clip", "use_fire_reset", "img_height", "img_width", "mode", "difficulty",
            "full_action_space", "repeat_action_probability", "use_inter_area_resize",
            "gray_scale", "num_threads")


def rom_path(conf: dict) -> str:
    """GetRomPath, atari_env.h:43-48."""
    return f"{conf['base_path']}/atari/roms/{conf['task']}.bin"


def _emulator_lib(conf: dict) -> str:
    return conf.get("emulator_lib") or os.environ.get("EPA_ATARI_EMULATOR_LIB", "")


def _atari_cfg(conf: dict, num_envs: int = 1, **kw: Any):
    params = {k: float(conf[k]) for k in _NUMERIC if k in conf}
    base, keep = native.make_config(num_envs, params=params, **kw)
    rom, lib = rom_path(conf).encode(), _emulator_lib(conf).encode()
    cfg = native.EpaAtariConfig(base, rom, lib)
    return cfg, (keep, rom, lib)


def _num_actions(conf: dict) -> int:
    """ActionSpec of the reference loads the ROM to size the action set (atari_env.h:76-90)."""
    cfg, keep = _atari_cfg(conf)
    n = ctypes.c_int32(0)
    native.check(native.lib().epa_atari_num_actions(ctypes.byref(cfg), ctypes.byref(n)))
    del keep
    return int(n.value)


class AtariDevicePool(DevicePool):
    """DevicePool over `epa_atari_create` (host emulator workers + HIP post-process)."""

    def __init__(self, conf: dict, **kw: Any) -> None:
        self._conf = conf
        kw.pop("params", None)
        super().__init__("Atari", conf["num_envs"], params={k: float(conf[k]) for k in _NUMERIC
                                                              if k in conf}, **kw)

    def _create(self, family: str, cfg: Any, params: Any) -> ctypes.c_void_p:
        rom, lib = rom_path(self._conf).encode(), _emulator_lib(self._conf).encode()
        acfg = native.EpaAtariConfig(cfg, rom, lib)
        h = ctypes.c_void_p()
        native.check(self._lib.epa_atari_create(ctypes.byref(acfg), ctypes.byref(h)))
        self.state_keys = native.pool_keys(h, "state")
        self.action_keys = native.pool_keys(h, "action")
        self.action_dtype = self.action_keys[-1][1]
        self.action_shape = self.action_keys[-1][2]
        return h


_Atari = FamilyDef(
    name="Atari", native="Atari",
    default_config=_ATARI_CONFIG,
    # StateSpec / ActionSpec, atari_env.h:64-90
    state_spec=lambda c: [
        ("obs", spec(np.uint8, [c["stack_num"] * (1 if c["gray_scale"] else 3),
                                c["img_height"], c["img_width"]], (0, 255))),
        ("info:lives", spec(np.int32, [-1])),
        ("info:reward", spec(np.float32, [-1])),
        ("info:terminated", spec(np.int32, [-1], (0, 1))),
        ("info:ram", spec(np.uint8, [128], (0, 255))),
    ],
    action_spec=lambda c: [("action", spec(np.int32, [-1], (0, _num_actions(c) - 1)))],
    pool_factory=lambda conf, kw: AtariDevicePool(conf, **kw),
)

_AtariEnvSpec, _AtariEnvPool = make_native_classes(_Atari, static_action_spec=[
    ("action", spec(np.int32, [-1]))])
AtariEnvSpec, AtariDMEnvPool, AtariGymnasiumEnvPool = py_env(_AtariEnvSpec, _AtariEnvPool)

__all__ = ["AtariEnvSpec", "AtariDMEnvPool", "AtariGymnasiumEnvPool", "AtariPostProcess"]


class AtariPostProcess:
    """frames [k, 2, 210, 160] u8 -> obs [k, stack_num (x3), img_height, img_width] u8."""

    def __init__(self, num_envs: int, stack_num: int = 4, img_height: int = 84,
                 img_width: int = 84, raw_height: int = 210, raw_width: int = 160,
                 use_inter_area_resize: bool = True, device: int = 0,
                 gray_scale: bool = True, palette: np.ndarray | None = None) -> None:
        """`palette`: frames are ALE palette indices and this is the table to apply on the
        device -- [256] u8 (gray_scale) or [3, 256] u8 planar RGB (gray_scale=False, which
        needs it); obs are then [k, stack_num * 3, h, w] (atari_env.h:320-335)."""
        self._lib = native.lib()
        self.num_envs, self.stack_num = num_envs, stack_num
        self.planes = stack_num * (1 if gray_scale else 3)
        self.out_hw = (img_height, img_width)
        self.raw_hw = (raw_height, raw_width)
        h = ctypes.c_void_p()
        pal = None
        if palette is not None:
            pal = np.ascontiguousarray(palette, dtype=np.uint8)
            assert pal.size == (256 if gray_scale else 768), pal.shape
        native.check(self._lib.epa_atari_post_create_ex(
            num_envs, stack_num, raw_height, raw_width, img_height, img_width,
            1 if use_inter_area_resize else 0, 1 if gray_scale else 0,
            pal.ctypes.data if pal is not None else None, device, ctypes.byref(h)))
        self._h = h
        self._frames = None  # pinned frame buffer handed to the emulator loop
        from envpool_amd.core.device_pool import _PinnedBlocks

        self._blocks = _PinnedBlocks(self._lib)

    def frame_buffer(self) -> np.ndarray:
        """Pinned host array [num_envs, 2, raw_h, raw_w] u8 for the emulator loop to write
        `maxpool_buf_[0/1]` into (atari_env.h:230-237): `push` then uploads straight from it
        with `hipMemcpyAsync`, chunk by chunk, overlapped with the kernel and the download of
        the previous chunk.  (Pageable arrays work too, at about 2/3 of the rate.)"""
        if self._frames is None:
            nb = self.num_envs * 2 * self.raw_hw[0] * self.raw_hw[1]
            ptr = self._lib.epa_host_alloc(nb)
            if not ptr:
                raise MemoryError(f"epa_host_alloc({nb}) failed")
            self._frames_ptr = ptr
            self._frames = np.ctypeslib.as_array((ctypes.c_ubyte * nb).from_address(ptr)).reshape(
                self.num_envs, 2, *self.raw_hw)
        return self._frames

    def push(self, frames: np.ndarray, env_id: np.ndarray | None = None,
             reset_mask: np.ndarray | None = None) -> np.ndarray:
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        k = frames.shape[0]
        assert frames.shape == (k, 2, *self.raw_hw), frames.shape
        ids = (np.arange(k, dtype=np.int32) if env_id is None
               else np.ascontiguousarray(env_id, dtype=np.int32))
        mask = None
        if reset_mask is not None:
            mask = np.ascontiguousarray(reset_mask, dtype=np.uint8)
        # observations land in a pinned block that the returned array owns (recycled when it
        # is garbage collected), like DevicePool.recv
        nb = k * self.planes * self.out_hw[0] * self.out_hw[1]
        obs = self._blocks.take(max(nb, 1))[:nb].reshape(k, self.planes, *self.out_hw)
        native.check(self._lib.epa_atari_post_push(
            self._h, ids.ctypes.data, k, frames.ctypes.data,
            mask.ctypes.data if mask is not None else None, obs.ctypes.data))
        return obs

    def push_device(self, d_frames: int, d_obs: int, k: int, d_env_id: int | None = None,
                    d_reset_mask: int | None = None) -> None:
        """Device-resident variant: raw device addresses in, nothing copied."""
        native.check(self._lib.epa_atari_post_push_device(
            self._h, ctypes.c_void_p(d_env_id), k, ctypes.c_void_p(d_frames),
            ctypes.c_void_p(d_reset_mask), ctypes.c_void_p(d_obs)))

    @property
    def stream(self) -> int:
        return int(self._lib.epa_atari_post_stream(self._h) or 0)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.epa_atari_post_destroy(self._h)  # drains the streams first
            self._h = None
        if getattr(self, "_frames", None) is not None:
            self._frames = None
            self._lib.epa_host_free(self._frames_ptr)
        if getattr(self, "_blocks", None) is not None:
            self._blocks.close()

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:
            pass
