import sys, time, json
sys.path.insert(0, '/root/repo')
import torch
from envpool_amd.core.device_pool import DevicePool
def run(n, b, streams, layout, steps=600):
    pool = DevicePool("HalfCheetah", n, batch_size=b, seed=0, max_episode_steps=1000, params={"compute_streams": streams, "planar_layout": layout})
    dev = torch.device("cuda", 0)
    ring = [torch.rand((b, 6), device=dev, dtype=torch.float64) * 2 - 1 for _ in range(8)]
    ids = torch.arange(n, device=dev, dtype=torch.int32); torch.cuda.synchronize()
    for j in range(n // b): pool.send_device(None, b, ids[j * b:].data_ptr())
    def cycle(i):
        ptrs, k = pool.recv_device(); pool.send_device(ring[i % 8].data_ptr(), k, ptrs[0])
    for i in range(4 * (n // b)): cycle(i)
    pool.synchronize(); t0 = time.perf_counter()
    for i in range(steps): cycle(i)
    pool.synchronize(); return b * steps / (time.perf_counter() - t0)
for n, b, s, l in ((65536, 32768, 2, 2), (65536, 16384, 4, 2), (65536, 8192, 4, 2), (65536, 8192, 8, 2), (65536, 21845, 3, 2), (32768, 16384, 2, 2), (16384, 8192, 2, 4), (16384, 8192, 2, 2)):
    print(json.dumps({"n": n, "b": b, "streams": s, "layout": l, "env_steps_per_s": run(n, b, s, l)}), flush=True)
