import sys, ctypes, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle.orc import Oracle
from mj_util import _H
N=64
o = Oracle("HumanoidStandup", N, seed=3, max_episode_steps=1000)
L=o.lib; inner = ctypes.cast(o.h, ctypes.POINTER(_H)).contents.h
vp=ctypes.c_void_p
L.mjcpu_raw_get.argtypes=[vp,ctypes.c_int,vp,vp,vp]
sc=np.zeros(256); L.mjcpu_model_scalars.argtypes=[vp,vp]; L.mjcpu_model_scalars(inner, sc.ctypes.data)
nq,nv=int(sc[0]),int(sc[1])
o.reset()
rng=np.random.default_rng(0)
rows=[];its=[]
for t in range(220):
    a=rng.uniform(-0.4,0.4,size=(N,17))
    o.step(a)
    if t%10==9 or t<5:
        r=[];i=[]
        for e in range(N):
            q=np.zeros(nq);v=np.zeros(nv);m=np.zeros(16)
            L.mjcpu_raw_get(inner,e,q.ctypes.data,v.ctypes.data,m.ctypes.data)
            r.append(int(m[3])); i.append(int(m[4]))
        r=np.array(r); i=np.array(i)
        print(t, "rows: min %d med %d p90 %d max %d | >24: %.2f >32: %.2f >40: %.2f >48 %.2f| iters med %d p90 %d max %d"%(r.min(),np.median(r),np.percentile(r,90),r.max(),(r>24).mean(),(r>32).mean(),(r>40).mean(),(r>48).mean(),np.median(i),np.percentile(i,90),i.max()), flush=True)
