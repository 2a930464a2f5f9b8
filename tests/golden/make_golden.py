"""Generate golden rollouts from the REFERENCE ITSELF (oracle/_ref = the
reference's C++ compiled in place from /root/reference; see oracle/Makefile).

Run in the build container only (needs /root/reference):
    make -C oracle ref && python tests/golden/make_golden.py
Writes tests/golden/<task_id>.npz holding the seeded action sequence and every
state key the reference returned for reset + T steps (auto-resets included).
The reference has no golden vectors of its own for these envs, so
these files are the pin for oracle/restate and for the HIP engine.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle.orc import Oracle  # noqa: E402
from oracle_cases import CASES, sample_actions  # noqa: E402

N, T, SEED = 8, 400, 42


def main() -> None:
    for name, c in CASES.items():
        o = Oracle(c["task"], N, seed=SEED, max_episode_steps=c["max_steps"],
                   extra=c["extra"], kind="reference", num_threads=2)
        rng = np.random.default_rng(1234)
        frames = [o.reset()]
        actions = []
        for _ in range(T):
            a = sample_actions(c, rng, N)
            actions.append(a)
            frames.append(o.step(a))
        out = {"actions": np.stack(actions), "seed": np.int64(SEED)}
        for k in frames[0]:
            out["state/" + k] = np.stack([f[k] for f in frames])
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
