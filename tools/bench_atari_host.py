"""Host path of the Atari post-process (frames on the host in, stacked observations
on the host out): PCIe-bound.  `push` pipelines upload / kernel / download over three
streams in chunks; frames written into the pinned `frame_buffer()` reach full rate."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from envpool_amd.atari import AtariPostProcess  # noqa: E402

n = 1024
post = AtariPostProcess(n)
rng = np.random.default_rng(0)
frames = rng.integers(0, 256, (n, 2, 210, 160), dtype=np.uint8)
ids = np.arange(n, dtype=np.int32)
pinned = post.frame_buffer()
pinned[:] = frames
for name, src in (("pageable frames", frames), ("pinned frame_buffer()", pinned)):
    for _ in range(3):
        post.push(src, ids, None)
    t0 = time.perf_counter()
    for _ in range(20):
        obs = post.push(src, ids, None)
    dt = (time.perf_counter() - t0) / 20
    print("%-22s %.2f ms/push  %.1f GB/s in+out  %.2fM env-pushes/s"
          % (name, dt * 1e3, (frames.nbytes + obs.nbytes) / dt / 1e9, n / dt / 1e6))
