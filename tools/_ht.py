import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from envpool_amd.core.device_pool import DevicePool
dev = torch.device("cuda", 0)
for task in ("Humanoid", "HumanoidStandup", "Ant"):
  for n in (64, 8192):
    pool = DevicePool(task, n, seed=0, max_episode_steps=1000)
    adim = int(np.prod(pool.action_shape))
    act = (torch.rand((n, adim), device=dev, dtype=torch.float64) * 2 - 1) * 0.4
    torch.cuda.synchronize()
    pool.send_device(None); pool.recv_device(); pool.synchronize()
    pool.set_timing(True)
    for i in range(5):
        pool.send_device(None); pool.recv_device()
    ms_r, l = pool.kernel_time_ms()
    pool.set_timing(False); pool.set_timing(True)
    for i in range(5):
        pool.send_device(act.data_ptr()); pool.recv_device()
    ms_s, l2 = pool.kernel_time_ms()
    print(task, n, "reset launch ms", ms_r, l, "step launch ms", ms_s, l2, "=> per forward (step) ms", ms_s / 20)
