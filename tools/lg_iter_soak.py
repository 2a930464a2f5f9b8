"""Soak of the one-evaluation line search (mj_planar_lg.hip.h::Solve) on the GPU: Newton trips per env-step of EVERY env
over many steps of the benchmark workload (random actions, auto-reset on), read back through get_state (slot 3 nv: the
trips the env's forward passes of the last env-step took, summed).  A forward pass that reached the exact-search
fallback (trip kLsExactAfter = 8) or the iteration cap (50) shows as a sum far outside the bulk.
usage: tools/lg_iter_soak.py [steps] [num_envs]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from envpool_amd.core.device_pool import DevicePool  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
for task, adim, nv, passes in (("HalfCheetah", 6, 9, 5), ("Walker2d", 6, 9, 16), ("Hopper", 3, 6, 16)):
    pool = DevicePool(task, n, seed=0, max_episode_steps=1000)
    ids = np.arange(n, dtype=np.int32)
    pool.reset(ids)
    pool.recv()
    rng = np.random.default_rng(1234)
    hist = np.zeros(1024, np.int64)
    worst = 0
    bad = 0
    for t in range(steps):
        pool.send(ids, rng.uniform(-1, 1, size=(n, adim)))
        out = pool.recv_dict()
        it = pool.get_state()[:, 3 * nv].astype(np.int64)
        live = out["elapsed_step"].ravel() > 0  # (a reset row keeps the count of the step before it)
        it = it[live]
        if it.size == 0:  # (every env reset in this launch: the common truncation of the HalfCheetah at step 1000)
            continue
        hist += np.bincount(np.minimum(it, 1023), minlength=1024)
        worst = max(worst, int(it.max()))
        bad += int(np.isnan(out["obs"]).any())
    tot = hist.sum()
    mean = (hist * np.arange(1024)).sum() / tot
    tail = {k: int(hist[k:].sum()) for k in (passes + 8, passes + 16, 50)}
    print(f"{task} N={n} {steps} steps ({tot:.3g} env-steps, {passes} forward passes each): trips per env-step mean "
          f"{mean:.2f}, max {worst}; env-steps with >= passes+8 / passes+16 / 50 trips: {tail}; NaN batches {bad}", flush=True)
    print("  histogram (trips: env-steps):", {int(k): int(v) for k, v in enumerate(hist) if v}, flush=True)
