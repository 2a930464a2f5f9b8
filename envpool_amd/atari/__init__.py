"""Atari observation post-process on the GPU (K4).

The reference's AtariEnv (envpool/atari/atari_env.h) runs ALE on the host and
then max-pools the last two frames, resizes to 84x84 (cv::resize INTER_AREA)
and pushes into a 4-deep frame stack (`PushStack`, atari_env.h:308-346).  ALE,
its ROMs and OpenCV are not part of this repository; this module exposes the
post-process as a batched HIP kernel behind the C ABI so that a host ALE loop
can hand over `maxpool_buf_[0/1]` of every env and get the stacked observation
back.  Gray-scale (the default `gray_scale=True`) only; INTER_AREA (the default
`use_inter_area_resize=True`) or INTER_LINEAR (`False`, the reference benchmark's setting).
"""

from __future__ import annotations

import ctypes

import numpy as np

from envpool_amd.core import native


class AtariPostProcess:
    """frames [k, 2, 210, 160] u8 -> obs [k, stack_num, img_height, img_width] u8."""

    def __init__(self, num_envs: int, stack_num: int = 4, img_height: int = 84,
                 img_width: int = 84, raw_height: int = 210, raw_width: int = 160,
                 use_inter_area_resize: bool = True, device: int = 0) -> None:
        self._lib = native.lib()
        self.num_envs, self.stack_num = num_envs, stack_num
        self.out_hw = (img_height, img_width)
        self.raw_hw = (raw_height, raw_width)
        h = ctypes.c_void_p()
        native.check(self._lib.epa_atari_post_create(
            num_envs, stack_num, raw_height, raw_width, img_height, img_width,
            1 if use_inter_area_resize else 0, device, ctypes.byref(h)))
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
        nb = k * self.stack_num * self.out_hw[0] * self.out_hw[1]
        obs = self._blocks.take(max(nb, 1))[:nb].reshape(k, self.stack_num, *self.out_hw)
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
