"""Long soak of the MuJoCo step kernels through the host path (numpy in / numpy out, auto-reset on): every observation and
reward of every env-step finite and bounded, elapsed_step inside the horizon, episodes ending where the task says.
A rare event of the solver (a forward pass at the iteration cap, a NaN that the healthy test must turn into a reset:
ant.h:214-229) shows here at rates the 150-1200-step soaks of the trip counts cannot see.
usage: tools/long_soak.py <task> <num_envs> <steps> <action dim> [action bound]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from envpool_amd.core.device_pool import DevicePool  # noqa: E402

task, n, steps, adim = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
hi = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
pool = DevicePool(task, n, seed=0, max_episode_steps=1000)
ids = np.arange(n, dtype=np.int32)
rng = np.random.default_rng(1234)
ring = [rng.uniform(-hi, hi, size=(n, adim)) for _ in range(64)]
pool.reset(ids)
pool.recv()
bad_obs = bad_rew = 0
omax = 0.0
rsum = 0.0
done_n = trunc_n = 0
emax = 0
t0 = time.perf_counter()
for t in range(steps):
    pool.send(ids, ring[(t * 7) % 64])
    out = pool.recv_dict()
    obs, rew = out["obs"], out["reward"]
    fo = np.isfinite(obs)
    if not fo.all():
        bad_obs += int((~fo).any(axis=tuple(range(1, obs.ndim))).sum())
    else:
        omax = max(omax, float(np.abs(obs).max()))
    fr = np.isfinite(rew)
    bad_rew += int((~fr).sum())
    rsum += float(rew[fr].sum())
    done_n += int(out["done"].sum())
    trunc_n += int(out["trunc"].sum())
    emax = max(emax, int(out["elapsed_step"].max()))
dt = time.perf_counter() - t0
print(f"{task} N={n} {steps} steps = {n * steps:.3g} env-steps in {dt:.0f} s: rows with a non-finite observation {bad_obs}, "
      f"non-finite rewards {bad_rew}, max |obs| {omax:.3g}, mean reward {rsum / (n * steps):.4f}, episodes ended {done_n} "
      f"(truncated {trunc_n}), max elapsed_step {emax}", flush=True)
assert bad_obs == 0 and bad_rew == 0 and emax <= 1000
