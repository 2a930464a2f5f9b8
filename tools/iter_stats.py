"""Newton-iteration statistics of the HalfCheetah kernel (diagnostic, GPU)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from envpool_amd.core.device_pool import DevicePool

n = 65536
pool = DevicePool("HalfCheetah", n, seed=0, max_episode_steps=1000, params={"precision": 1})
ids = np.arange(n, dtype=np.int32)
pool.reset(ids); pool.recv()
rng = np.random.default_rng(1234)
for t in range(120):
    act = rng.uniform(-1, 1, size=(n, 6))
    pool.send(ids, act); pool.recv()
    if t in (20, 60, 119):
        it = pool.get_state()[:, 27]
        w = it.reshape(-1, 64)
        wm = w.max(axis=1)
        print(f"step {t}: iters/env-step mean {it.mean():.2f} p50 {np.median(it):.0f} p99 {np.percentile(it,99):.0f} "
              f"p99.9 {np.percentile(it,99.9):.0f} max {it.max():.0f} | per-wave max: mean {wm.mean():.1f} "
              f"p50 {np.median(wm):.0f} p99 {np.percentile(wm,99):.0f} max {wm.max():.0f}")
