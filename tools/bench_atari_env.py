"""Atari end to end (BASELINE config 5: num_envs=1024, host emulator step + HIP post-process)
through the reference-compatible host API: send(actions) / recv() -> numpy.  ALE is not
available offline: the emulator is the synthetic console of tests/synth_ale, which is far cheaper
than ALE (~1 us per frame instead of ~150 us), so this measures the ENGINE -- worker pool, row
claiming, pinned staging, H2D + palette/max-pool/resize/stack kernel + D2H -- not emulation.
usage: python tools/bench_atari_env.py [num_envs] [steps] [gray_scale] [num_threads]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from atari_util import plugin_path  # noqa: E402
from envpool_amd.atari import AtariDevicePool  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
gray = int(sys.argv[3]) if len(sys.argv) > 3 else 1
# the synthetic console costs ~1 us per frame: an eighth of the hardware threads serves it best (0 = the
# reference's default, min(batch_size, hardware_concurrency), the right rule for a real emulator)
threads = int(sys.argv[4]) if len(sys.argv) > 4 else max(1, (os.cpu_count() or 8) // 8)
conf = dict(num_envs=n, task="synth_fire", base_path="/synthetic", emulator_lib=plugin_path(),
            stack_num=4, frame_skip=4, noop_max=30, gray_scale=gray, use_inter_area_resize=0,
            img_height=84, img_width=84, num_threads=threads)
pool = AtariDevicePool(conf, seed=0, max_episode_steps=27000)
ids = np.arange(n, dtype=np.int32)
rng = np.random.default_rng(0)
acts = [rng.integers(0, 6, n).astype(np.int32) for _ in range(8)]
pool.reset(ids)
pool.recv()
for i in range(10):
    pool.send(ids, acts[i % 8])
    pool.recv()
t0 = time.perf_counter()
for i in range(steps):
    pool.send(ids, acts[i % 8])
    out = pool.recv()
dt = time.perf_counter() - t0
obs_bytes = out[8].nbytes
print(json.dumps({"task": "Atari (synthetic console)", "num_envs": n, "steps": steps,
                  "gray_scale": bool(gray), "ms_per_step": 1e3 * dt / steps,
                  "env_steps_per_s": n * steps / dt, "frames_per_s": 4 * n * steps / dt,
                  "obs_MB_per_step": obs_bytes / 1e6,
                  "pcie_GBps_in_plus_out": (n * 2 * 33600 + obs_bytes) * steps / dt / 1e9}))
pool.close()
