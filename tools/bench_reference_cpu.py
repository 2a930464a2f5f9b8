"""The reference's own AsyncEnvPool (compiled in place into oracle/_ref by
`make -C oracle ref`; see oracle/ref_driver.cc) timed on this box's host cores:
the CPU baseline of SURVEY §8(d) for the classic_control / toy_text families.
Sync step loop from C++ (Send / Recv), num_threads = all cores.  Note: the shim
semaphore is mutex + condvar, the reference's moodycamel one spins."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.orc import Oracle, have_ref  # noqa: E402

if not have_ref():
    sys.exit("oracle/_ref/libref_oracle.so missing (built only where /root/reference exists)")
cores = os.cpu_count() or 1
# family, max_episode_steps, extra config (tests/oracle_cases.py), num_envs, steps
for task, max_steps, extra, n, steps in (
        ("CartPole", 500, (), 64, 20000), ("CartPole", 500, (), 65536, 60),
        ("Pendulum", 200, (1,), 65536, 60), ("Acrobot", 500, (), 65536, 40),
        ("FrozenLake", 100, (4,), 65536, 60)):
    for threads in sorted({min(cores, 8), cores}):
        o = Oracle(task, n, seed=0, max_episode_steps=max_steps, extra=extra, kind="reference",
                   num_threads=threads)
        o.reset()
        rng = np.random.default_rng(0)
        if o.action_dtype == np.int32:
            act = rng.integers(0, 2, size=(n, o.action_elems)).astype(np.int32)
        else:
            act = rng.uniform(-1, 1, size=(n, o.action_elems)).astype(o.action_dtype)
        o.time_steps(max(2, steps // 10), act)
        t = o.time_steps(steps, act)
        print(json.dumps({"task": task, "num_envs": n, "threads": threads, "host_cores": cores,
                          "env_steps_per_s": n * steps / t, "kind": "reference"}))
        o.close()
