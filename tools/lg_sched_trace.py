"""Who ran which chunk when, inside ONE launch of the lane-group planar kernel (diagnostic build
-DEPA_LG_SCHED_TRACE: tools/build_alt_lg.sh sched -mllvm -disable-machine-licm -mllvm -amdgpu-spill-sgpr-to-vgpr=false
-DEPA_LG_SCHED_TRACE, copied over libenvpool_amd.so on the GPU box).  Every chunk files {wave, chunk, start, end} in
100 MHz wall-clock ticks; from them: the launch's span, the waves' busy fraction, chunk-duration statistics, the
makespan an ideal (perfectly balanced) and a clairvoyant LPT schedule of the same chunks would have.

    python tools/lg_sched_trace.py [task] [num_envs] [launches] [key=value ...]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from envpool_amd.core import native
from envpool_amd.core.device_pool import DevicePool


def lpt_makespan(dur, slots):
    import heapq
    h = [0.0] * slots
    for d in sorted(dur, reverse=True):
        heapq.heappush(h, heapq.heappop(h) + d)
    return max(h)


def main():
    task = sys.argv[1] if len(sys.argv) > 1 else "HalfCheetah"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    launches = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    params = {"precision": 1}
    for kv in sys.argv[4:]:
        k, v = kv.split("=")
        params[k] = float(v)
    lib = native.lib()
    f = lib.epa_debug_lg_sched
    f.restype = ctypes.c_longlong
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    pool = DevicePool(task, n, seed=0, max_episode_steps=1000, params=params)
    adim = int(np.prod(pool.action_shape))
    ring = [torch.rand((n, adim), device="cuda", dtype=torch.float64) * 2 - 1 for _ in range(8)]
    pool.send_device(None)
    pool.recv_device()
    for i in range(300):
        pool.send_device(ring[i % 8].data_ptr())
        pool.recv_device()
    pool.synchronize()
    cap = 1 << 15
    buf = np.zeros(4 * cap + 4, np.uint64)  # + the reset-branch counters behind the records
    assert f(None, 0, 1) >= 0
    rows = []
    for i in range(launches):
        pool.send_device(ring[i % 8].data_ptr())
        pool.recv_device()
        pool.synchronize()
        k = f(buf.ctypes.data, cap, 1)
        assert 0 < k <= cap, k
        rst = buf[4 * cap:4 * cap + 4].astype(np.float64)
        r = buf[:4 * k].reshape(k, 4).astype(np.int64)
        t0 = r[:, 2].min()
        start, end = (r[:, 2] - t0) * 0.01, (r[:, 3] - t0) * 0.01  # us
        dur = end - start
        waves = len(np.unique(r[:, 0]))
        span = end.max()
        busy = dur.sum() / (waves * span)
        per_wave_end = np.array([end[r[:, 0] == w].max() for w in np.unique(r[:, 0])])
        per_wave_n = np.bincount(r[:, 0].astype(int))
        rows.append(dict(chunks=k, waves=waves, span=span, busy=busy, mean=dur.mean(), std=dur.std(),
                         p5=np.percentile(dur, 5), p50=np.median(dur), p95=np.percentile(dur, 95), mx=dur.max(),
                         ideal=dur.sum() / waves, lpt=lpt_makespan(dur, waves),
                         first_idle=per_wave_end.min(), end_p50=np.median(per_wave_end),
                         n1=int((per_wave_n == 1).sum()), n2=int((per_wave_n == 2).sum()), n3=int((per_wave_n >= 3).sum()),
                         start_last=start.max(),
                         reset_branch_us=rst[0] * 0.01 / max(rst[1], 1), chunks_with_reset=rst[2] / max(rst[1], 1),
                         mj_steps_us=rst[3] * 0.01 / max(rst[1], 1)))
    print(f"{task} N={n} params={params}: {launches} launches (after 300 warm-up steps), per launch:")
    keys = list(rows[0])
    for k in keys:
        v = np.array([r[k] for r in rows], float)
        print(f"  {k:12s} mean {v.mean():10.2f}   min {v.min():10.2f}   max {v.max():10.2f}")
    print("  (us; span = first chunk start .. last chunk end; busy = sum of chunk durations / (waves x span); ideal = sum / "
          "waves; lpt = clairvoyant longest-first list schedule of the same chunks; first_idle / end_p50 = when the first / "
          "the median wave ran out of work; n1 n2 n3 = waves that ran 1 / 2 / >= 3 chunks; start_last = start of the last chunk; reset_branch_us = mean time per chunk between its top and the start of its stepping branch, i.e. the reset "
          "branch the wave runs first for the lanes whose env resets; chunks_with_reset = fraction of chunks in which some env reset; mj_steps_us = mean time per chunk inside the "
          "frame_skip x mj_step loop incl. the state loads: chunk mean - mj_steps_us = reset branch + stores + bookkeeping)")


if __name__ == "__main__":
    main()
