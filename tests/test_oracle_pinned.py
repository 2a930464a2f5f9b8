"""Pin the plain-C oracle (oracle/restate) before anything trusts it.

(a) bit-for-bit against the golden rollouts generated from the reference's own
    C++ (tests/golden/*.npz, made by tests/golden/make_golden.py);
(b) bit-for-bit against oracle/_ref (the reference compiled in place) whenever
    that library is present (build container only).
"""
import os

import numpy as np
import pytest

from oracle.orc import Oracle, have_ref
from oracle_cases import CASES, sample_actions

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def replay_golden(make_pool, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    acts = g["actions"]
    n = acts.shape[1]
    pool = make_pool(n, int(g["seed"]))
    frames = [pool.reset()]
    for t in range(acts.shape[0]):
        frames.append(pool.step(acts[t]))
    for key in [k for k in g.files if k.startswith("state/")]:
        want = g[key]
        got = np.stack([f[key[6:]] for f in frames]).reshape(want.shape)
        yield key[6:], want, got


@pytest.mark.parametrize("name", sorted(CASES))
def test_port_matches_golden(name):
    c = CASES[name]

    def make(n, seed):
        return Oracle(c["task"], n, seed=seed, max_episode_steps=c["max_steps"],
                      extra=c["extra"], kind="port")

    for key, want, got in replay_golden(make, name):
        assert got.dtype == want.dtype, key
        assert np.array_equal(got, want), f"{name}:{key} differs from reference"


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built here")
@pytest.mark.parametrize("name", sorted(CASES))
def test_port_matches_compiled_reference(name):
    c = CASES[name]
    n = 96
    ref = Oracle(c["task"], n, seed=7, max_episode_steps=c["max_steps"],
                 extra=c["extra"], kind="reference", num_threads=2)
    port = Oracle(c["task"], n, seed=7, max_episode_steps=c["max_steps"],
                  extra=c["extra"], kind="port")
    assert [k[:2] for k in ref.keys] == [k[:2] for k in port.keys]
    ra, rb = ref.reset(), port.reset()
    rng = np.random.default_rng(99)
    for t in range(700):
        for k in ra:
            assert np.array_equal(ra[k], rb[k]), (name, t, k)
        a = sample_actions(c, rng, n)
        ra, rb = ref.step(a), port.step(a)


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built here")
def test_partial_env_id_step_matches_reference():
    """sync mode, subset of env ids: rows come back in send order
    (envpool/core/state_buffer.h:94-97, async_envpool.h:170-175)."""
    c = CASES["CartPole-v1"]
    n = 16
    ref = Oracle("CartPole", n, seed=1, max_episode_steps=500, kind="reference",
                 num_threads=2)
    port = Oracle("CartPole", n, seed=1, max_episode_steps=500, kind="port")
    ref.reset(), port.reset()
    ids = np.array([5, 2, 11, 7], dtype=np.int32)
    rng = np.random.default_rng(0)
    for _ in range(50):
        a = sample_actions(c, rng, len(ids))
        ra, rb = ref.step(a, ids), port.step(a, ids)
        for k in ra:
            assert np.array_equal(ra[k], rb[k]), k
        assert np.array_equal(ra["info:env_id"].ravel(), ids)


def test_atari_fixtures_come_from_the_reference_compiled_in_place():
    """tests/golden/atari_*.npz must be what oracle/_ref/libref_atari.so -- the reference's own
    atari_env.h over the synthetic console -- produces (regenerate with
    tests/golden/make_atari_golden.py); also checks that the emulator plugin the product loads
    and the reference's ALE shim wrap the same console."""
    import zlib

    import atari_cases as ac
    from oracle import orc

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    if not orc.have_ref_atari():
        pytest.skip("oracle/_ref/libref_atari.so not built (no /root/reference here)")
    for name, (_, n, seed, max_steps, steps) in {**ac.CASES, **ac.BIG_CASES}.items():
        g = np.load(os.path.join(ROOT, "tests", "golden", f"atari_{name}.npz"))
        c = ac.config(name)
        o = orc.Oracle("Atari", n, seed=seed, max_episode_steps=max_steps, extra=ac.extra(c),
                       kind="reference_atari", num_threads=2)
        b = o.reset()
        for t in range(min(steps, 40) + 1):
            np.testing.assert_array_equal(ac.crc_rows(b["obs"]), g["obs_crc"][t])
            np.testing.assert_array_equal(b["reward"].ravel(), g["reward"][t])
            np.testing.assert_array_equal(b["elapsed_step"].ravel(), g["elapsed_step"][t])
            b = o.step(g["actions"][t])
        assert zlib.crc32(b["obs"].tobytes()) != 0


def test_resize_matches_opencv():
    """Activates when tools/pin_with_opencv.py has been run somewhere with OpenCV: the plain-C
    restatement of cv::resize (oracle/atari/atari_post.c) against real cv2 outputs, gray and
    per channel of 3-channel images."""
    import ctypes

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "opencv_resize.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/opencv_resize.npz absent (OpenCV not installable offline: PARITY "
                    "UNPINNED for cv::resize; run tools/pin_with_opencv.py where cv2 imports)")
    from oracle import orc

    L = orc._load(orc.PORT_LIB)
    g = np.load(path)
    src = g["src"]
    for key in g.files:
        if "_" not in key or key in ("src", "cv_version"):
            continue
        kind, mode, hw = key.split("_")
        h, w = (int(x) for x in hw.split("x"))
        fn = L.orc_resize_area_u8 if mode == "area" else L.orc_resize_linear_u8
        for i in range(src.shape[0]):
            chans = [0] if kind == "gray" else [0, 1, 2]
            for c in chans:
                plane = np.ascontiguousarray(src[i, :, :, c])
                out = np.zeros((h, w), dtype=np.uint8)
                fn(plane.ctypes.data_as(ctypes.c_void_p), 210, 160, out.ctypes.data_as(ctypes.c_void_p), h, w)
                want = g[key][i] if kind == "gray" else g[key][i][:, :, c]
                np.testing.assert_array_equal(out, want, err_msg=f"{key} frame {i} channel {c}")
