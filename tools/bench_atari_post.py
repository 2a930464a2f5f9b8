"""Atari post-process kernel rate (device-resident frames in, observations out): INTER_AREA
(the reference's default) and INTER_LINEAR (its benchmark's setting)."""
import sys

import torch

sys.path.insert(0, ".")
from envpool_amd.atari import AtariPostProcess  # noqa: E402

dev = torch.device("cuda", 0)
for area in (True, False):
    for n in (1024, 16384):
        post = AtariPostProcess(n, use_inter_area_resize=area)
        frames = torch.randint(0, 256, (n, 2, 210, 160), device=dev, dtype=torch.uint8)
        obs = torch.empty((n, 4, 84, 84), device=dev, dtype=torch.uint8)
        torch.cuda.synchronize()
        stream = torch.cuda.ExternalStream(post.stream, device=dev)
        for _ in range(5):
            post.push_device(frames.data_ptr(), obs.data_ptr(), n)
        stream.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            t0.record()
            for _ in range(50):
                post.push_device(frames.data_ptr(), obs.data_ptr(), n)
            t1.record()
        stream.synchronize()
        ms = t0.elapsed_time(t1) / 50
        alg = 2 * 33600 + 3 * 7056 + 7056 + 4 * 7056
        print("INTER_AREA" if area else "INTER_LINEAR", n, "%.1f us" % (ms * 1e3),
              "%.0f GB/s" % (alg * n / (ms * 1e-3) / 1e9))
