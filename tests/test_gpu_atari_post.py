"""GPU parity of the Atari post-process kernel vs oracle/atari/atari_post.c
(bit-exact: u8 outputs).  The oracle itself is UNPINNED against OpenCV 4.13
(not installed; the reference only tests shapes, image_process_test.cc:23-40)."""
import ctypes

import numpy as np
import pytest

from envpool_amd.atari import AtariPostProcess
from oracle.orc import PORT_LIB

pytestmark = pytest.mark.gpu


class OraclePost:
    def __init__(self, n, s=4, oh=84, ow=84, linear=False):
        self.L = ctypes.CDLL(PORT_LIB)
        self.L.orc_atari_post_create.restype = ctypes.c_void_p
        self.L.orc_atari_post_push.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 3
        self.h = ctypes.c_void_p(self.L.orc_atari_post_create(n, s, 210, 160, oh, ow))
        self.L.orc_atari_post_set_linear.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self.L.orc_atari_post_set_linear(self.h, int(linear))
        self.s, self.oh, self.ow = s, oh, ow

    def push(self, frames, ids, mask):
        k = len(ids)
        obs = np.zeros((k, self.s, self.oh, self.ow), np.uint8)
        self.L.orc_atari_post_push(self.h, ids.ctypes.data, k, frames.ctypes.data,
                                   mask.ctypes.data if mask is not None else None,
                                   obs.ctypes.data)
        return obs


def pong_like(rng, k):
    """Sparse Atari-like frames: background + a few bright rectangles."""
    f = np.full((k, 2, 210, 160), 87, np.uint8)
    for i in range(k):
        for j in range(2):
            for _ in range(6):
                y, x = rng.integers(0, 200), rng.integers(0, 150)
                f[i, j, y:y + rng.integers(2, 16), x:x + rng.integers(1, 8)] = rng.integers(0, 256)
    return f


def test_post_process_bit_exact_with_resets_and_partial_ids():
    n = 64
    gpu, orc = AtariPostProcess(n), OraclePost(n)
    rng = np.random.default_rng(0)
    ids = np.arange(n, dtype=np.int32)
    frames = pong_like(rng, n)
    mask = np.ones(n, np.uint8)  # reset: replicate into every stack slot
    a, b = gpu.push(frames, ids, mask), orc.push(frames, ids, mask)
    np.testing.assert_array_equal(a, b)
    assert (a[:, 0] == a[:, 3]).all()
    for t in range(12):
        if t % 3 == 2:  # partial, shuffled ids with some resets
            sub = rng.permutation(n)[:17].astype(np.int32)
            frames = rng.integers(0, 256, (17, 2, 210, 160), dtype=np.uint8)
            mask = (rng.random(17) < 0.3).astype(np.uint8)
        else:
            sub, frames, mask = ids, pong_like(rng, n), None
        a, b = gpu.push(frames, sub, mask), orc.push(frames, sub, mask)
        np.testing.assert_array_equal(a, b, err_msg=f"push {t}")
    # stack ordering: newest frame last, previous newest moved to slot 2
    prev = gpu.push(pong_like(rng, n), ids, None)
    frames = pong_like(rng, n)
    a = gpu.push(frames, ids, None)
    np.testing.assert_array_equal(a[:, :3], prev[:, 1:])


def test_post_process_at_baseline_config5_size():
    """BASELINE.json config 5: num_envs = 1024 (the size `tools/bench_families.py` times AtariPostKernel at):
    full-batch pushes incl. the reset push, a partial shuffled push and a push with scattered resets, bit-exact."""
    n = 1024
    gpu, orc = AtariPostProcess(n), OraclePost(n)
    rng = np.random.default_rng(5)
    ids = np.arange(n, dtype=np.int32)
    for t in range(6):
        if t == 3:
            sub = rng.permutation(n)[:333].astype(np.int32)
            frames, mask = pong_like(rng, 333), (rng.random(333) < 0.3).astype(np.uint8)
        else:
            sub = ids
            frames = pong_like(rng, n) if t % 2 == 0 else rng.integers(0, 256, (n, 2, 210, 160), dtype=np.uint8)
            mask = np.ones(n, np.uint8) if t == 0 else ((rng.random(n) < 0.05).astype(np.uint8) if t == 5 else None)
        a, b = gpu.push(frames, sub, mask), orc.push(frames, sub, mask)
        assert a.shape == (len(sub), 4, 84, 84)
        np.testing.assert_array_equal(a, b, err_msg=f"push {t}")


@pytest.mark.parametrize("oh,ow", [(64, 64), (96, 75), (40, 30), (100, 150)])
def test_post_process_other_sizes(oh, ow):
    """Non-default img_height / img_width: more taps per pixel than the 84x84
    specialisation allows (generic <6,6> kernel), widths that are not a multiple
    of 4 (byte-store path), upscaling-free mixes."""
    n = 16
    gpu, orc = AtariPostProcess(n, img_height=oh, img_width=ow), OraclePost(n, oh=oh, ow=ow)
    rng = np.random.default_rng(1)
    ids = np.arange(n, dtype=np.int32)
    for t in range(5):
        frames = rng.integers(0, 256, (n, 2, 210, 160), dtype=np.uint8)
        mask = np.ones(n, np.uint8) if t == 0 else (rng.random(n) < 0.2).astype(np.uint8)
        a, b = gpu.push(frames, ids, mask), orc.push(frames, ids, mask)
        assert a.shape == (n, 4, oh, ow)
        np.testing.assert_array_equal(a, b, err_msg=f"push {t}")


@pytest.mark.parametrize("oh,ow", [(84, 84), (64, 64), (96, 75), (40, 30)])
def test_post_process_bilinear(oh, ow):
    """use_inter_area_resize=False: cv::INTER_LINEAR's 8-bit fixed-point path (what the
    reference's benchmark selects, benchmark/test_envpool.py:92), bit-exact vs the oracle,
    incl. resets, partial ids and sizes off the 84x84 default."""
    n = 32
    gpu = AtariPostProcess(n, img_height=oh, img_width=ow, use_inter_area_resize=False)
    orc = OraclePost(n, oh=oh, ow=ow, linear=True)
    rng = np.random.default_rng(2)
    ids = np.arange(n, dtype=np.int32)
    for t in range(8):
        if t % 3 == 2:
            sub = rng.permutation(n)[:11].astype(np.int32)
            frames = rng.integers(0, 256, (11, 2, 210, 160), dtype=np.uint8)
            mask = (rng.random(11) < 0.3).astype(np.uint8)
        else:
            sub = ids
            frames = pong_like(rng, n) if t % 2 else rng.integers(0, 256, (n, 2, 210, 160), dtype=np.uint8)
            mask = np.ones(n, np.uint8) if t == 0 else None
        a, b = gpu.push(frames, sub, mask), orc.push(frames, sub, mask)
        np.testing.assert_array_equal(a, b, err_msg=f"push {t}")
    # a constant image stays constant; a horizontal ramp stays monotone
    flat = np.full((n, 2, 210, 160), 131, np.uint8)
    assert (gpu.push(flat, ids, np.ones(n, np.uint8)) == 131).all()
    ramp = np.broadcast_to(np.arange(160, dtype=np.uint8)[None, None, None, :], (n, 2, 210, 160))
    out = gpu.push(np.ascontiguousarray(ramp), ids, np.ones(n, np.uint8))[:, -1].astype(int)
    assert (np.diff(out, axis=2) >= 0).all()


def test_post_process_errors():
    with pytest.raises(ValueError):
        AtariPostProcess(4, img_height=105, img_width=80)  # integer scale: fast path
    with pytest.raises(ValueError):  # INTER_LINEAR with an exact 2x2 reduction: area-fast path
        AtariPostProcess(4, img_height=105, img_width=80, use_inter_area_resize=False)


def test_post_matches_opencv_golden():
    """GPU leg of the cv::resize pin (tools/pin_with_opencv.py): the HIP post-process against real
    cv2 outputs -- gray frames, and RGB through an identity-per-channel palette."""
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "opencv_resize.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/opencv_resize.npz absent (OpenCV not installable offline)")
    from envpool_amd.atari import AtariPostProcess

    g = np.load(path)
    src = g["src"]
    n = src.shape[0]
    for key in g.files:
        if "_" not in key or key in ("src", "cv_version"):
            continue
        kind, mode, hw = key.split("_")
        h, w = (int(x) for x in hw.split("x"))
        if kind == "gray":
            post = AtariPostProcess(n, stack_num=1, img_height=h, img_width=w,
                                    use_inter_area_resize=mode == "area")
            fr = np.stack([src[:, :, :, 0], src[:, :, :, 0]], axis=1)
            obs = post.push(fr, reset_mask=np.ones(n, dtype=np.uint8))
            np.testing.assert_array_equal(obs[:, 0], g[key], err_msg=key)
            post.close()
        else:  # three planes from one index frame: push each channel as "indices" with an
            # identity palette for that plane and zeros elsewhere is equivalent to the RGB path
            for c in range(3):
                pal = np.zeros((3, 256), dtype=np.uint8)
                pal[c] = np.arange(256, dtype=np.uint8)
                post = AtariPostProcess(n, stack_num=1, img_height=h, img_width=w,
                                        use_inter_area_resize=mode == "area", gray_scale=False,
                                        palette=pal)
                fr = np.stack([src[:, :, :, c], src[:, :, :, c]], axis=1)
                obs = post.push(fr, reset_mask=np.ones(n, dtype=np.uint8))
                np.testing.assert_array_equal(obs[:, c], g[key][:, :, :, c], err_msg=f"{key} c{c}")
                post.close()
