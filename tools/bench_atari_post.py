import sys, torch
sys.path.insert(0,'.')
from envpool_amd.atari import AtariPostProcess
dev=torch.device('cuda',0)
for n in (1024,16384):
    post=AtariPostProcess(n)
    frames=torch.randint(0,256,(n,2,210,160),device=dev,dtype=torch.uint8)
    obs=torch.empty((n,4,84,84),device=dev,dtype=torch.uint8)
    torch.cuda.synchronize()
    stream=torch.cuda.ExternalStream(post.stream,device=dev)
    for _ in range(5): post.push_device(frames.data_ptr(),obs.data_ptr(),n)
    stream.synchronize()
    t0,t1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        t0.record()
        for _ in range(50): post.push_device(frames.data_ptr(),obs.data_ptr(),n)
        t1.record()
    stream.synchronize()
    ms=t0.elapsed_time(t1)/50; alg=2*33600+3*7056+7056+4*7056
    print(n, "%.1f us"%(ms*1e3), "%.0f GB/s"%(alg*n/(ms*1e-3)/1e9))
