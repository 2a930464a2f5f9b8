import sys, time, ctypes, numpy as np
sys.path.insert(0,'.')
from envpool_amd.atari import AtariPostProcess
from envpool_amd.core import native
L=native.lib()
n=1024
post=AtariPostProcess(n)
rng=np.random.default_rng(0)
frames=rng.integers(0,256,(n,2,210,160),dtype=np.uint8)
ids=np.arange(n,dtype=np.int32)
for _ in range(3): post.push(frames,ids,None)
t0=time.perf_counter()
for _ in range(20): post.push(frames,ids,None)
dt=(time.perf_counter()-t0)/20
print("pageable in/out: %.2f ms/push  %.1f GB/s in+out"%(dt*1e3,(frames.nbytes+n*4*84*84)/dt/1e9))
# pinned input and output
nb=frames.nbytes; p=L.epa_host_alloc(nb); pin=np.ctypeslib.as_array((ctypes.c_ubyte*nb).from_address(p)).reshape(frames.shape); pin[:]=frames
ob=n*4*84*84; po=L.epa_host_alloc(ob); pobs=np.ctypeslib.as_array((ctypes.c_ubyte*ob).from_address(po)).reshape(n,4,84,84)
def push_pinned():
    native.check(L.epa_atari_post_push(post._h, ids.ctypes.data, n, pin.ctypes.data, None, pobs.ctypes.data))
for _ in range(3): push_pinned()
t0=time.perf_counter()
for _ in range(20): push_pinned()
dt=(time.perf_counter()-t0)/20
print("pinned in/out:   %.2f ms/push  %.1f GB/s in+out"%(dt*1e3,(frames.nbytes+ob)/dt/1e9))
