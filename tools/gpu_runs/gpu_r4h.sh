#!/bin/bash
# round 4, call h: Humanoid pools with per-stream scratch: Humanoid tests, async A/B (compute_streams 1 vs 4) of
# Humanoid / HumanoidStandup with 4 batches of 16384 in flight, device path
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_api.py tests/test_gpu_mujoco.py tests/test_gpu_fullsize.py tests/test_gpu_mujoco_golden.py tests/test_gpu_device_path.py -q -m gpu -x -k "umanoid or async" 2>&1 | tail -6 > gpurun_out/r4h_tests.log
cat gpurun_out/r4h_tests.log
python - <<'PY' > gpurun_out/r4h_humanoid_async_streams.txt 2>&1
import time, numpy as np, torch
from envpool_amd.core.device_pool import DevicePool
dev = torch.device("cuda", 0)
for task, amp in (("Humanoid", 0.4), ("HumanoidStandup", 0.4)):
    n, b = 65536, 16384
    for streams in (1, 4):
        pool = DevicePool(task, n, batch_size=b, seed=0, max_episode_steps=1000, params={"compute_streams": streams})
        ring = [(torch.rand((b, 17), device=dev, dtype=torch.float64) * 2 - 1) * amp for _ in range(8)]
        ids = torch.arange(n, device=dev, dtype=torch.int32)
        for j in range(n // b):
            pool.send_device(None, b, ids[j * b:].data_ptr())
        def cycle(i):
            ptrs, k = pool.recv_device()
            pool.send_device(ring[i % 8].data_ptr(), k, ptrs[0])
        for i in range(24): cycle(i)
        pool.synchronize(); t = time.perf_counter()
        K = 80 if task == "Humanoid" else 32
        for i in range(K): cycle(i)
        pool.synchronize(); t = time.perf_counter() - t
        print(task, "4 x 16384 in flight, compute_streams", streams, "%.4g env-steps/s" % (b * K / t), "%.2f ms per batch" % (1e3 * t / K), flush=True)
        del pool
PY
cat gpurun_out/r4h_humanoid_async_streams.txt
