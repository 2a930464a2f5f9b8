#!/bin/bash
# Round 6, second session: direct step (epa_send_into: the step kernel writes its rows into the caller's pinned block)
export TMPDIR=/tmp
O=gpurun_out/r6n; mkdir -p $O
( time timeout 1800 python -m pytest tests/test_gpu_api.py tests/test_gpu_device_path.py tests/test_gpu_step_pipeline.py tests/test_gpu_blocking_recv.py tests/test_gpu_lifecycle.py tests/test_gpu_sharded.py -m gpu -q -x ) > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=|FAILED|Error" $O/tests.log | tail -6
for rep in 1 2; do for cfg in "HalfCheetah 65536 6" "Walker2d 65536 6" "Hopper 65536 3" "Ant 32768 8" "Pusher 65536 7" "Humanoid 16384 17" "HalfCheetah 8192 6" "HalfCheetah 32768 6" "HalfCheetah 131072 6"; do
  set -- $cfg
  for d in 0 1; do
    echo "direct_out=$d rep$rep $(EPA_PARAMS=direct_out=$d python tools/numpy_step_ab.py $1 $2 -1 $3 bind 2>/dev/null | tail -1 | sed 's/.*bound.: True} //')"
  done
done; done | tee $O/direct_out_ab.txt
