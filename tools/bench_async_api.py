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


def run_async_device(task, n, batch, steps, streams, adim=6):
    """Device path: recv_device -> send_device with the env ids and actions resident in HBM, `n / batch`
    batches in flight, successive batches on `streams` compute streams (engine key compute_streams)."""
    import torch

    from envpool_amd.core.device_pool import DevicePool

    fam = task.split("-")[0]
    pool = DevicePool(fam, n, batch_size=batch, seed=0, max_episode_steps=1000, params={"compute_streams": streams})
    dev = torch.device("cuda", 0)
    ring = [torch.rand((batch, adim), device=dev, dtype=torch.float64) * 2 - 1 for _ in range(8)]
    torch.cuda.synchronize()
    ids = torch.arange(n, device=dev, dtype=torch.int32)
    torch.cuda.synchronize()
    for j in range(n // batch):  # n / batch reset launches = the batches in flight
        pool.send_device(None, batch, ids[j * batch:].data_ptr())

    def cycle(i):
        ptrs, k = pool.recv_device()
        pool.send_device(ring[i % 8].data_ptr(), k, ptrs[0])  # ptrs[0] = info:env_id of the batch
    for i in range(4 * (n // batch)):
        cycle(i)
    pool.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        cycle(i)
    pool.synchronize()
    return batch * steps / (time.perf_counter() - t0)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "streams":
        # independent in-flight batches on several compute streams (round 3): 8 batches of 8192 in flight
        # then fewer, larger batches: two 32768-env batches in flight fill each other's tail
        for task, n, b, adim in (("HalfCheetah-v4", 65536, 8192, 6), ("Ant-v4", 65536, 8192, 8),
                                 ("HalfCheetah-v4", 65536, 16384, 6), ("HalfCheetah-v4", 65536, 32768, 6),
                                 ("HalfCheetah-v4", 131072, 65536, 6)):
            rec = {"task": task, "num_envs": n, "batch_size": b}
            for k in (1, 2, 4, 8):
                rec[f"device_path_{k}_streams_env_steps_per_s"] = run_async_device(task, n, b, 400 if "Half" in task else 100, k, adim)
            print(json.dumps(rec), flush=True)
        sys.exit(0)
    for task, b in (("HalfCheetah-v4", 65536), ("HalfCheetah-v4", 8192), ("Ant-v4", 32768)):
        steps = 100 if task.startswith("Half") else 20
        rec = {"task": task, "batch_size": b,
               "sync_env_steps_per_s": run_sync(task, b, steps),
               "async_2_batches_env_steps_per_s": run_async(task, 2 * b, b, steps),
               "async_3_batches_env_steps_per_s": run_async(task, 3 * b, b, steps)}
        print(json.dumps(rec))
