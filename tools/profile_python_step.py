"""Where the host time of a small-batch step() goes (cProfile of the Python adaptor + ctypes path):
python tools/profile_python_step.py [task] [num_envs] [steps]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import envpool_amd as envpool  # noqa: E402

task = sys.argv[1] if len(sys.argv) > 1 else "CartPole-v1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
env = envpool.make(task, "gymnasium", num_envs=n, seed=0)
env.reset()
sp = env.action_space
rng = np.random.default_rng(0)
if hasattr(sp, "n"):
    acts = [rng.integers(0, sp.n, n).astype(np.int32) for _ in range(8)]
else:
    acts = [rng.uniform(-1, 1, (n, *sp.shape)).astype(sp.dtype) for _ in range(8)]
for i in range(200):
    env.step(acts[i % 8])
t0 = time.perf_counter()
for i in range(steps):
    env.step(acts[i % 8])
dt = time.perf_counter() - t0
print(f"{task} N={n}: {dt / steps * 1e6:.1f} us per step() = {n * steps / dt:.3e} env-steps/s")
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    env.step(acts[i % 8])
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
