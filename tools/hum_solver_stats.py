"""Per-env solver statistics of the quad Humanoid kernel ("hum_debug" & 16: the info keys carry row
visits, sweeps, the wave's rows and streaming solves of the env-step) -- run on a GPU box."""
import sys, numpy as np
sys.path.insert(0,'/root/repo')
from envpool_amd.core.device_pool import DevicePool
n=4096
pool=DevicePool("Humanoid",n,seed=0,max_episode_steps=1000,params={"hum_debug":16})
ids=np.arange(n,dtype=np.int32); pool.reset(ids); pool.recv_dict()
rng=np.random.default_rng(0)
for t in range(40):
    pool.send(ids, rng.uniform(-1,1,(n,17))); d=pool.recv_dict()
    if t in (10,20,30,39):
        vis=d["info:x_position"].ravel(); sw=d["info:y_position"].ravel(); rows=d["info:distance_from_origin"].ravel(); st=d["info:x_velocity"].ravel()
        live=d["elapsed_step"].ravel()>0
        w=slice(0,None,16)
        print(t,"per env-step (20 forwards): visits mean %.0f max %.0f; sweeps mean %.0f; wave rows sum mean %.1f; streaming solves mean %.2f; resets %.2f"%(vis[live].mean(),vis.max(),sw[live].mean(),rows[live].mean(),st[live].mean(),1-live.mean()))
