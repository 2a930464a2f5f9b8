#!/bin/bash
# Round 6, second session: long soak of the MuJoCo kernels (tools/long_soak.py)
export TMPDIR=/tmp
O=gpurun_out/r6s; mkdir -p $O
{
timeout 600 python tools/long_soak.py HalfCheetah 65536 20000 6
timeout 600 python tools/long_soak.py Walker2d 65536 12000 6
timeout 600 python tools/long_soak.py Hopper 65536 12000 3
timeout 600 python tools/long_soak.py Ant 32768 8000 8
timeout 600 python tools/long_soak.py Pusher 65536 6000 7 2
timeout 600 python tools/long_soak.py Humanoid 16384 1500 17 0.4
timeout 600 python tools/long_soak.py HumanoidStandup 16384 600 17 0.4
timeout 300 python tools/long_soak.py Swimmer 65536 6000 2
timeout 300 python tools/long_soak.py InvertedDoublePendulum 65536 6000 1
timeout 300 python tools/long_soak.py Reacher 65536 6000 2
} 2>&1 | grep -v amdgpu.ids | tee $O/long_soak.txt
