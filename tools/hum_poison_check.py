"""DIAGNOSTIC: run-to-run determinism of the MuJoCo kernels with every CU's LDS filled with NaN bit
patterns before each step (tools/lds_poison): a kernel that reads an LDS slot before writing it shows
up as NaN / differing output on every run instead of once in a while.
usage: python tools/hum_poison_check.py [task ...]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from envpool_amd.core.device_pool import DevicePool  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "tools", "lds_poison", "libpoison.so"))
lib.poison_lds.argtypes = [ctypes.c_void_p, ctypes.c_uint]
TASKS = {"Humanoid": (17, {"post_constraint": 1, "use_contact_force": 1}), "HumanoidStandup": (17, {}),
         "Ant": (8, {}), "HalfCheetah": (6, {}), "Walker2d": (6, {}), "Hopper": (3, {}), "Pusher": (7, {}),
         "Swimmer": (2, {}), "Reacher": (2, {}), "InvertedDoublePendulum": (1, {})}
for task in sys.argv[1:] or list(TASKS):
    adim, params = TASKS[task]
    n = 256
    bad = 0
    for trial in range(3):
        pools = [DevicePool(task, n, seed=5, max_episode_steps=1000, params=params) for _ in range(2)]
        ids = np.arange(n, dtype=np.int32)
        rng = np.random.default_rng(1)
        outs = []
        for p in pools:
            p.reset(ids)
            outs.append(p.recv_dict())
        for t in range(40):
            a = rng.uniform(-1, 1, (n, adim))
            outs = []
            for i, p in enumerate(pools):
                if i == 1 or trial > 0:  # pool 0 of trial 0 runs unpoisoned
                    assert lib.poison_lds(ctypes.c_void_p(p.stream), 0xFFFFFFFF if (t + trial) % 2 else 0x7FF80001) == 0
                p.send(ids, a)
                outs.append(p.recv_dict())
            nan = [int(np.isnan(o["obs"]).sum()) for o in outs]
            same = np.array_equal(outs[0]["obs"], outs[1]["obs"], equal_nan=True)
            if nan[0] or nan[1] or not same:
                bad += 1
                if bad <= 3:
                    print(f"  {task} trial {trial} step {t}: nan counts {nan}, identical {same}")
    print(f"{task}: {'OK' if bad == 0 else f'{bad} bad steps'}")
