"""A/B of the host path's sync step (send(numpy) + recv() -> numpy) with and without the two-launch pipeline
(engine key "step_pipeline"):  python tools/numpy_step_ab.py <task> <num_envs> <step_pipeline rows, 0 = off> <action dim>
[bind: the process on the CPUs of the GPU's NUMA node first]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import numpy as np
from envpool_amd.core.device_pool import DevicePool
if len(sys.argv) > 5 and sys.argv[5] == "bind":
    from envpool_amd.core.affinity import bind_host_to_device
    print(bind_host_to_device(0), end=" ")
task = sys.argv[1]; n = int(sys.argv[2]); sp = int(sys.argv[3]); adim = int(sys.argv[4])
extra = {k: float(v) for k, v in (kv.split("=") for kv in os.environ.get("EPA_PARAMS", "").split(",") if kv)}  # more engine keys
pool = DevicePool(task, n, seed=0, max_episode_steps=1000, params={**({"step_pipeline": sp} if sp >= 0 else {}), **extra})
ids = np.arange(n, dtype=np.int32); rng = np.random.default_rng(0)
hact = [rng.uniform(-1, 1, size=(n, adim)) for _ in range(4)]
pool.reset(ids); pool.recv()
for i in range(20): pool.send(ids, hact[i % 4]); pool.recv()
t = time.perf_counter()
for i in range(200): pool.send(ids, hact[i % 4]); pool.recv()
dt = time.perf_counter() - t
ts = 0.0
for i in range(100):
    t0 = time.perf_counter(); pool.send(ids, hact[i % 4]); ts += time.perf_counter() - t0; pool.recv()
print(task, n, "step_pipeline", sp, "ms/step %.4f" % (dt / 200 * 1e3), "env-steps/s %.3e" % (n * 200 / dt), "send ms %.4f" % (ts * 10))
