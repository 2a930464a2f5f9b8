"""What the host link gives, next to what the numpy API of a 65536-env HalfCheetah step moves (review item:
15.2 MB down + 3.4 MB up in 0.68 ms = 22 GB/s -- is that the link?).

(a) hipMemcpyAsync between a pinned host buffer and HBM (torch pinned tensor, non_blocking copy = one
    hipMemcpyAsync), sizes 3 MB .. 256 MB, both directions, HIP events on the copy stream;
(b) the legs of one numpy-API step of the pool, serialised as the sync API serialises them: wall time of
    `send(numpy)` + `recv()` against the kernel time of the same launch (HIP events of the pool), so that
    (step - kernel) / bytes is the rate the two copies actually got, including their enqueue latencies.
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def link(nbytes, direction, reps=20):
    host = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    src, dst = (host, dev) if direction == "h2d" else (dev, host)
    for _ in range(3):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        dst.copy_(src, non_blocking=True)
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / reps
    # one copy, host clock: enqueue + DMA + completion wait, what a sync API pays per step
    w = []
    for _ in range(reps):
        a = time.perf_counter()
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        w.append(time.perf_counter() - a)
    return {"bytes": nbytes, "dir": direction, "back_to_back_ms": ms, "GBps": nbytes / ms / 1e6,
            "single_copy_wall_ms": 1e3 * float(np.median(w)), "single_copy_GBps": nbytes / float(np.median(w)) / 1e9}


def main():
    for nbytes in (3 << 20, 15_200_000, 64 << 20, 256 << 20):
        for d in ("h2d", "d2h"):
            print(json.dumps(link(nbytes, d)))
    from envpool_amd.core.device_pool import DevicePool

    n = 65536
    pool = DevicePool("HalfCheetah", n, seed=0, max_episode_steps=1000, params={"precision": 1})
    ids = np.arange(n, dtype=np.int32)
    rng = np.random.default_rng(0)
    act = [rng.uniform(-1, 1, size=(n, 6)) for _ in range(4)]
    pool.send(ids, act[0])
    out = pool.recv()
    down = sum(int(np.asarray(a).nbytes) for a in out)
    up = act[0].nbytes
    for i in range(5):
        pool.send(ids, act[i % 4])
        pool.recv()
    k = 100
    pool.set_timing(True)
    ts = tr = 0.0
    t0 = time.perf_counter()
    for i in range(k):
        a = time.perf_counter()
        pool.send(ids, act[i % 4])
        b = time.perf_counter()
        pool.recv()
        c = time.perf_counter()
        ts += b - a
        tr += c - b
    wall = time.perf_counter() - t0
    kernel_ms, launches = pool.kernel_time_ms()
    step_ms = 1e3 * wall / k
    print(json.dumps({"numpy_api_step_ms": step_ms, "send_call_ms": 1e3 * ts / k, "recv_call_ms": 1e3 * tr / k,
                      "kernel_ms": kernel_ms, "launches": launches, "bytes_down": down, "bytes_up": up,
                      "copies_ms": step_ms - kernel_ms,
                      "copies_GBps": (down + up) / ((step_ms - kernel_ms) * 1e-3) / 1e9,
                      "whole_step_GBps": (down + up) / (step_ms * 1e-3) / 1e9,
                      "env_steps_per_s": n / (step_ms * 1e-3)}))


if __name__ == "__main__":
    main()
