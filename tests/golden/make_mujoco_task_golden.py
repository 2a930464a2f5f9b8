"""TEST INFRASTRUCTURE.  Writes tests/golden/mujoco_task_<id>.npz from the reference's OWN
gym-MuJoCo task wrappers running inside the reference's own AsyncEnvPool
(oracle/_ref/libref_mujoco.so: envpool/mujoco/gym/*.h compiled in place, see
oracle/ref_mujoco_driver.cc) -- engine underneath = oracle/mjcpu (parity unpinned, mjcpu.h).
Run here (needs /root/reference for `make -C oracle ref`):

    python tests/golden/make_mujoco_task_golden.py

Each file: seed, max_episode_steps, extra, actions[T, n, nu] and, per state key, the array
[T + 1, n, elems] (index 0 = the reset).  Readers: tests/test_mjcpu_task_golden.py (the plain-C
restatement, bit for bit, CPU) and tests/test_gpu_mujoco_golden.py (the HIP path, through the
C ABI, to the stated tolerance)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.orc import Oracle  # noqa: E402
from tests.mj_util import GYM_VARIANTS, mj_extra  # noqa: E402

N, SEED = 3, 20260924


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, (task, max_steps, over) in sorted(GYM_VARIANTS.items()):
        steps = 24 if task.startswith("Humanoid") else 60
        extra = mj_extra(task, **over)
        ref = Oracle(task, N, seed=SEED, max_episode_steps=max_steps, extra=extra,
                     kind="reference_mujoco", num_threads=2)
        rng = np.random.default_rng(SEED)
        acts = rng.uniform(-1.0, 1.0, size=(steps, N, ref.action_elems))
        rows = [ref.reset()]
        for t in range(steps):
            rows.append(ref.step(acts[t]))
        data = {"key:" + k: np.stack([r[k] for r in rows]) for k in rows[0]}
        np.savez_compressed(os.path.join(out_dir, f"mujoco_task_{name}.npz"), seed=SEED,
                            max_episode_steps=max_steps, extra=np.asarray(extra), actions=acts,
                            task=task, **data)
        print(name, steps, "steps", sum(v.nbytes for v in data.values()) // 1024, "KiB raw")


if __name__ == "__main__":
    main()
