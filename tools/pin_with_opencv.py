"""Dumps cv2.resize golden outputs for the Atari post-process (the analogue of
tools/pin_with_mujoco.py).  OpenCV (4.13.0 in the reference: envpool/workspace0.bzl) is not
installable offline; run this wherever `import cv2` works:

    python tools/pin_with_opencv.py     ->  tests/golden/opencv_resize.npz

tests/test_oracle_pinned.py::test_resize_matches_opencv (CPU: oracle/atari/atari_post.c) and
tests/test_gpu_atari_post.py::test_post_matches_opencv_golden (HIP kernel) activate when the
file exists.  Inputs are seeded synthetic 210x160 frames (uint8, one and three channels) and
the (height, width) targets the tests use; INTER_AREA is the reference default
(use_inter_area_resize=True, image_process.h:31-32), INTER_LINEAR the `False` branch (:34)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = [(84, 84), (64, 96), (100, 100), (105, 80)]


def frames(seed=0, n=6):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        f = rng.integers(0, 256, size=(210, 160, 3), dtype=np.uint8)
        if i % 2:  # blocky, Atari-like content
            f = np.repeat(np.repeat(f[::7, ::8], 7, axis=0), 8, axis=1)[:210, :160]
        out.append(f)
    return np.stack(out)


def main():
    import cv2

    src = frames()
    out = {"src": src, "cv_version": np.array(cv2.__version__)}
    for (h, w) in SIZES:
        for mode, flag in (("area", cv2.INTER_AREA), ("linear", cv2.INTER_LINEAR)):
            if mode == "area" and 210 % h == 0 and 160 % w == 0:
                continue  # integer factors take resizeAreaFast, which the product rejects
            out[f"gray_{mode}_{h}x{w}"] = np.stack(
                [cv2.resize(np.ascontiguousarray(f[:, :, 0]), (w, h), interpolation=flag) for f in src])
            out[f"rgb_{mode}_{h}x{w}"] = np.stack(
                [cv2.resize(f, (w, h), interpolation=flag) for f in src])
    path = os.path.join(ROOT, "tests", "golden", "opencv_resize.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "OpenCV", cv2.__version__)


if __name__ == "__main__":
    try:
        main()
    except ImportError:
        sys.exit("cv2 is not importable here: nothing written (the tests stay skipped)")
