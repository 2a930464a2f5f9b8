"""Host (numpy) API with several batches in flight: envpool's async mode
(batch_size < num_envs; `recv()` then `send(actions, env_id)`), which exists to
overlap env stepping with the consumer.  Here it overlaps the PCIe copies of one
batch with the step kernel of the next (separate upload / kernel / download
streams in epa::Pool).  Compared with the sync `step()` loop on the same envs."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import envpool_amd as envpool  # noqa: E402


def run_sync(task, n, steps):
    env = envpool.make(task, "gymnasium", num_envs=n, seed=0)
    env.reset()
    rng = np.random.default_rng(0)
    acts = [rng.uniform(-1, 1, (n, *env.action_space.shape)) for _ in range(4)]
    for i in range(5):
        env.step(acts[i % 4])
    t0 = time.perf_counter()
    for i in range(steps):
        env.step(acts[i % 4])
    dt = time.perf_counter() - t0
    return n * steps / dt


def run_async(task, n, batch, steps):
    env = envpool.make(task, "gymnasium", num_envs=n, batch_size=batch, seed=0)
    env.async_reset()
    rng = np.random.default_rng(0)
    acts = [rng.uniform(-1, 1, (batch, *env.action_space.shape)) for _ in range(4)]
    for i in range(6):
        out = env.recv()
        env.send(acts[i % 4], out[-1]["env_id"])
    t0 = time.perf_counter()
    for i in range(steps):
        out = env.recv()
        env.send(acts[i % 4], out[-1]["env_id"])
    dt = time.perf_counter() - t0
    return batch * steps / dt


if __name__ == "__main__":
    for task, b in (("HalfCheetah-v4", 65536), ("HalfCheetah-v4", 8192), ("Ant-v4", 32768)):
        steps = 100 if task.startswith("Half") else 20
        rec = {"task": task, "batch_size": b,
               "sync_env_steps_per_s": run_sync(task, b, steps),
               "async_2_batches_env_steps_per_s": run_async(task, 2 * b, b, steps),
               "async_3_batches_env_steps_per_s": run_async(task, 3 * b, b, steps)}
        print(json.dumps(rec))
