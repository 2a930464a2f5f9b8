"""Per-env solver statistics of the quad Humanoid kernel ("hum_debug" & 16: the info keys carry row
visits, sweeps, the wave's rows and hybrid / streaming solves of the env-step) -- run on a GPU box with the
diagnostic build.   usage: tools/hum_solver_stats.py [Humanoid|HumanoidStandup] [num_envs]"""
import sys, numpy as np
sys.path.insert(0,'/root/repo')
from envpool_amd.core.device_pool import DevicePool
task=sys.argv[1] if len(sys.argv)>1 else "Humanoid"
n=int(sys.argv[2]) if len(sys.argv)>2 else 4096
pool=DevicePool(task,n,seed=0,max_episode_steps=1000,params={"hum_debug":16})
ids=np.arange(n,dtype=np.int32); pool.reset(ids); pool.recv_dict()
rng=np.random.default_rng(0)
keys=(["info:reward_linup","info:reward_quadctrl","info:reward_alive","info:reward_impact"] if task=="HumanoidStandup"
      else ["info:x_position","info:y_position","info:distance_from_origin","info:x_velocity"])
for t in range(60):
    pool.send(ids, rng.uniform(-0.4,0.4,(n,17))); d=pool.recv_dict()
    if t in (1,5,10,20,30,40,59):
        vis,sw,rows,st=(d[k].ravel() for k in keys)
        live=d["elapsed_step"].ravel()>0
        print(t,"per env-step (20 forwards): wave visits mean %.0f max %.0f; wave sweeps mean %.0f; wave rows sum (register solves) mean %.1f; hybrid solves mean %.2f (of 20); resets %.2f"%(vis[live].mean(),vis.max(),sw[live].mean(),rows[live].mean(),st[live].mean(),1-live.mean()), flush=True)
# stage timers (diagnostic build): cycles / 16 per env-step, wave level
names={32:["position+detection","smooth dynamics","rows","staging (register form)"],
       64:["sweeps (register form)","solver epilogue","streaming solver","forwards"],
       128:["solves started over","sweeps (hybrid)","staging (hybrid)","register-form solves"],
       256:["hybrid solves","wave rows, hybrid (sum)","register visits in hybrid sweeps","-"]}
for dbg in (32,64,128,256):
    pool=DevicePool(task,n,seed=0,max_episode_steps=1000,params={"hum_debug":dbg})
    pool.reset(ids); pool.recv_dict()
    rng=np.random.default_rng(0)
    acc=np.zeros(4); cnt=0
    for t in range(40):
        pool.send(ids, rng.uniform(-0.4,0.4,(n,17))); d=pool.recv_dict()
        if t>=20:
            acc+=np.array([d[k].ravel().mean() for k in keys]); cnt+=1
    for nm,v in zip(names[dbg],acc/cnt): print("%-22s %12.0f  (x16 cycles per env-step)"%(nm,v*1.0), flush=True)
