#!/bin/bash
# round 3: finite termination inside the line search for the Ant quad kernel: parity + bench
set -u
export TMPDIR=/tmp
O=gpurun_out/r3m
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_api.py tests/test_gpu_fullsize.py tests/test_gpu_sharded.py -m gpu -q -x -k "ant or Ant" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=|FAILED" $O/tests.log | tail -8
for n in 32768 65536; do for pr in fp64 fp32; do
  timeout 300 python bench.py --task Ant --num-envs $n --precision $pr --no-cpu-baseline 2>>$O/err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('Ant n=$n $pr %.4e ms/step %.4f'%(d['value'],d['ms_per_step']))" | tee -a $O/ant.txt
done; done
