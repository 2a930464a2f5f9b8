#!/bin/bash
# direct_out = 0 / 1 / 2 per family (numpy step, pool defaults otherwise)
export TMPDIR=/tmp
O=gpurun_out/r6n; mkdir -p $O
for rep in 1 2; do for cfg in "HalfCheetah 65536 6" "HalfCheetah 8192 6" "HalfCheetah 32768 6" "HalfCheetah 131072 6" "Walker2d 65536 6" "Hopper 65536 3" "Ant 32768 8" "Ant 65536 8" "Pusher 65536 7" "Humanoid 16384 17" "HumanoidStandup 8192 17" "InvertedDoublePendulum 65536 1" "Swimmer 65536 2" "Reacher 65536 2"; do
  set -- $cfg
  for d in 0 1 2; do
    echo "direct_out=$d rep$rep $(EPA_PARAMS=direct_out=$d timeout 120 python tools/numpy_step_ab.py $1 $2 -1 $3 bind 2>/dev/null | tail -1 | sed 's/.*bound.: True} //')"
  done
done; done | tee $O/direct_out2_ab.txt
