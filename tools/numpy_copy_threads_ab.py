"""Sweep of the engine key "copy_threads" (helper threads that stage the action rows of a pipelined sync step) on the
host path's step (send(numpy) + recv() -> numpy):  python tools/numpy_copy_threads_ab.py [task] [num_envs] [action dim]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import numpy as np
from envpool_amd.core.device_pool import DevicePool
task = sys.argv[1] if len(sys.argv) > 1 else "HalfCheetah"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
adim = int(sys.argv[3]) if len(sys.argv) > 3 else 6
ids = np.arange(n, dtype=np.int32); rng = np.random.default_rng(0)
hact = [rng.uniform(-1, 1, size=(n, adim)) for _ in range(4)]
for rep in range(3):
    for ct in (0, 1, 2, 3, 4, 6, 8):
        pool = DevicePool(task, n, seed=0, max_episode_steps=1000, params={"copy_threads": ct})
        pool.reset(ids); pool.recv()
        for i in range(30): pool.send(ids, hact[i % 4]); pool.recv()
        t = time.perf_counter()
        for i in range(300): pool.send(ids, hact[i % 4]); pool.recv()
        dt = time.perf_counter() - t
        ts = 0.0
        for i in range(100):
            t0 = time.perf_counter(); pool.send(ids, hact[i % 4]); ts += time.perf_counter() - t0; pool.recv()
        print(task, n, "copy_threads", ct, "rep", rep, "ms/step %.4f" % (dt / 300 * 1e3), "env-steps/s %.3e" % (n * 300 / dt),
              "send ms %.4f" % (ts * 10), flush=True)
        pool.close()
