"""Where a wave of the lane-group planar kernel spends its cycles (diagnostic build -DEPA_LG_TIMERS:
tools/build_alt_lg.sh lgtimers ... -DEPA_LG_TIMERS, copied over libenvpool_amd.so on the GPU box).

    python tools/lg_stage_timers.py [task] [num_envs] [steps] [planar_layout]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from envpool_amd.core import native
from envpool_amd.core.device_pool import DevicePool

CATS = ["load / store / loop overhead (RK4: + stage updates)", "kinematics + smooth forces + constraint rows",
        "row pass + group sums + stop tests", "factor / solve / M products", "line search",
        "Euler integration (implicit damping)"]


def main():
    task = sys.argv[1] if len(sys.argv) > 1 else "HalfCheetah"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    lib = native.lib()
    f = lib.epa_debug_lg_timers
    f.argtypes = [ctypes.c_void_p, ctypes.c_int]
    params = {"precision": 1}
    if len(sys.argv) > 4:
        params["planar_layout"] = int(sys.argv[4])
    pool = DevicePool(task, n, seed=0, max_episode_steps=1000, params=params)
    adim = int(np.prod(pool.action_shape))
    ring = [torch.rand((n, adim), device="cuda", dtype=torch.float64) * 2 - 1 for _ in range(8)]
    pool.send_device(None)
    pool.recv_device()
    for i in range(60):
        pool.send_device(ring[i % 8].data_ptr())
        pool.recv_device()
    pool.synchronize()
    out = np.zeros(16, np.uint64)
    assert f(out.ctypes.data, 1) == 0
    for i in range(steps):
        pool.send_device(ring[i % 8].data_ptr())
        pool.recv_device()
    pool.synchronize()
    assert f(out.ctypes.data, 1) == 0
    cyc = np.concatenate([out[:5], out[10:11]]).astype(np.float64)
    chunks = float(out[8])
    tot = cyc.sum()
    print(f"{task} N={n} {params}: {steps} launches, {chunks / steps:.0f} chunks per launch, "
          f"{float(out[9]) / chunks:.0f} cycles per chunk (100 MHz s_memtime ticks x core ratio: relative only)")
    for name, c in zip(CATS, cyc):
        print(f"  {name:48s} {100 * c / tot:5.1f} %   {c / chunks:9.0f} ticks / chunk")
    print(f"  per chunk (= env-step of a wave): forward passes {out[7] / chunks:.2f}, Newton trips {out[5] / chunks:.2f}, "
          f"line-search evaluations {out[6] / chunks:.2f}")
    solver = cyc[2] + cyc[3] + cyc[4]
    print(f"  solver share of the wave's cycles: {100 * solver / tot:.1f} %")


if __name__ == "__main__":
    main()
