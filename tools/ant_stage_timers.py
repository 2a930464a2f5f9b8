"""Where a wave of the Ant kernel spends its cycles, and how well the unit queue packs the launch (diagnostic build
-DEPA_ANT_TIMERS: tools/build_ant_timers.sh -> lib/libenvpool_amd_anttimers.so, picked up through ENVPOOL_AMD_LIB).

    ENVPOOL_AMD_LIB=envpool_amd/lib/libenvpool_amd_anttimers.so python tools/ant_stage_timers.py [num_envs] [steps] [key=value ...]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from envpool_amd.core import native
from envpool_amd.core.device_pool import DevicePool

CATS = ["unit overhead: ticket, state loads / stores, outputs", "front end: kinematics, inertias, smooth forces, limit rows",
        "contact set-up", "pass over the rows + quad sums + stop tests", "factor / solve", "line search (M s + evaluation)",
        "RK4 stage updates + position integration", "waiting for the chunk's previous unit"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    params = {"precision": 1}
    for kv in sys.argv[3:]:
        k, v = kv.split("=")
        params[k] = float(v)
    lib = native.lib()
    f = lib.epa_debug_ant_timers
    f.argtypes = [ctypes.c_void_p, ctypes.c_int]
    warm_arg = params.pop("warmup", 2000)
    pool = DevicePool("Ant", n, seed=0, max_episode_steps=1000, params=params)
    params["warmup"] = warm_arg
    ring = [torch.rand((n, 8), device="cuda", dtype=torch.float64) * 2 - 1 for _ in range(8)]
    pool.send_device(None)
    pool.recv_device()
    warm = int(params.pop("warmup", 2000))  # steady state: episodes end and restart (bench.py's window)
    for i in range(warm):
        pool.send_device(ring[i % 8].data_ptr())
        pool.recv_device()
    pool.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    assert f(buf, 1) == 0
    pool.set_timing(2)
    for i in range(steps):
        pool.send_device(ring[i % 8].data_ptr())
        pool.recv_device()
    ms, launches = pool.kernel_time_ms()
    assert f(buf, 1) == 0
    t = np.array(list(buf), dtype=np.float64)
    cyc, trips, passes, units, life, waves = t[:8], t[8], t[9], t[10], t[11], t[12]
    tot = cyc.sum()
    print(f"sphere classes per forward pass: the wave's union {t[13] / passes:.2f} (what the class loops run over), "
          f"the busiest lane's own set {t[14] / passes:.2f}")
    print(f"Ant N={n} params={params}: {launches} launches, {ms:.4f} ms per launch = {n / ms * 1e3:.3e} env-steps/s")
    print(f"per launch: {waves / launches:.0f} waves, {units / launches:.0f} units, {passes / launches / (n / 16):.1f} forward "
          f"passes and {trips / launches / (n / 16):.1f} Newton trips per chunk and env-step")
    print("| stage | share of a wave's cycles | kclk per chunk and env-step |")
    print("|---|---|---|")
    for name, c in zip(CATS, cyc):
        print(f"| {name} | {100 * c / tot:.1f} % | {c / launches / (n / 16) / 1e3:.1f} |")
    # packing: the launch lasts as long as its longest wave; busy = everything but waiting, idle = launch - life
    clk_per_ms = life / waves / ms  # a wave lives (almost) the whole launch
    busy = (tot - cyc[7]) / waves
    print(f"mean wave: life {life / waves / 1e3:.0f} kclk (~{clk_per_ms / 1e3:.0f} kclk per ms if it spans the launch), busy "
          f"{busy / 1e3:.0f} kclk = {100 * busy / (life / waves):.1f} % of its life, waiting {100 * cyc[7] / tot:.2f} %")


if __name__ == "__main__":
    main()
