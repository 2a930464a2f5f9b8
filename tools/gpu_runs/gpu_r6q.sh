#!/bin/bash
# Round 6, second session: tiny host-path batches without DMA commands (engine key small_zero_copy): suite + A/B
export TMPDIR=/tmp
O=gpurun_out/r6q; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q -x ) > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; grep -E "passed|failed|rc=|FAILED|real" $O/gpu_tests.log | tail -6
cat > /tmp/ab.py <<'PY'
import sys, time, numpy as np
sys.path.insert(0, ".")
from envpool_amd.core.device_pool import DevicePool
for rep in range(3):
    for task, n in (("CartPole", 64), ("CartPole", 1024), ("Pendulum", 256), ("HalfCheetah", 64), ("Ant", 64), ("FrozenLake", 4096)):
        for zc in (0, 1):
            pool = DevicePool(task, n, seed=0, max_episode_steps=200, params={"small_zero_copy": zc})
            ids = np.arange(n, dtype=np.int32)
            rng = np.random.default_rng(0)
            if np.issubdtype(pool.action_dtype, np.integer):
                acts = [rng.integers(0, 2, (n, *pool.action_shape)).astype(pool.action_dtype) for _ in range(8)]
            else:
                acts = [rng.uniform(-1, 1, (n, *pool.action_shape)).astype(pool.action_dtype) for _ in range(8)]
            pool.reset(ids); pool.recv()
            for i in range(300): pool.send(ids, acts[i % 8]); pool.recv()
            t0 = time.perf_counter()
            for i in range(3000): pool.send(ids, acts[i % 8]); pool.recv()
            dt = time.perf_counter() - t0
            print(f"{task} N={n} small_zero_copy={zc} rep{rep}: {dt / 3000 * 1e6:.1f} us per send + recv", flush=True)
            pool.close()
PY
python /tmp/ab.py 2>&1 | grep -v amdgpu.ids | tee $O/small_zero_copy_ab.txt
