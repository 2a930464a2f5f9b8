"""PCIe-inclusive rate of the reference-compatible numpy API (T_numpy_api of
SURVEY §8d): envpool_amd.make(...).step(numpy actions) -> numpy outputs, i.e.
H2D of the action batch + kernel + D2H of every state key per step.  This is
NOT the headline `value` (which keeps inputs/outputs resident in HBM); it is
quoted in DESIGN.md §5."""
import json
import sys
import os
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import envpool_amd as envpool  # noqa: E402


def run(task, n, steps, adim=None):
    env = envpool.make(task, "gymnasium", num_envs=n, seed=0)
    env.reset()
    rng = np.random.default_rng(0)
    sp = env.action_space
    if hasattr(sp, "n"):
        acts = [rng.integers(0, sp.n, n).astype(np.int32) for _ in range(8)]
    else:
        acts = [rng.uniform(-1, 1, (n, *sp.shape)).astype(sp.dtype) for _ in range(8)]
    for i in range(5):
        env.step(acts[i % 8])
    t0 = time.perf_counter()
    for i in range(steps):
        env.step(acts[i % 8])
    dt = time.perf_counter() - t0
    # send/recv split: overlap the next send with nothing (sync API), so also
    # report the pipelined pattern send(t+1) before recv(t) is not legal here.
    return {"task": task, "num_envs": n, "steps": steps, "ms_per_step": 1e3 * dt / steps,
            "env_steps_per_s": n * steps / dt}


if __name__ == "__main__":
    if len(sys.argv) > 1:  # e.g. Humanoid-v4 65536 20
        print(json.dumps(run(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))))
        sys.exit(0)
    for task, n, steps in (("HalfCheetah-v4", 65536, 100), ("HalfCheetah-v4", 8192, 300),
                           ("Walker2d-v4", 65536, 100),
                           ("Ant-v4", 32768, 20), ("CartPole-v1", 65536, 200),
                           ("CartPole-v1", 64, 2000), ("FrozenLake-v1", 65536, 200)):
        print(json.dumps(run(task, n, steps)))
