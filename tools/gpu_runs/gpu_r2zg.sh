#!/bin/bash
# Pusher with the spread launch: parity + A/B at N = 32768 / 16384 / 65536
set -u
export TMPDIR=/tmp
O=gpurun_out/r2zg
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_sharded.py -m gpu -q -k "usher or planar_spread" > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log; tail -3 $O/tests.log
for n in 16384 32768 65536; do for sp in 1 0; do
  timeout 300 python bench.py --task Pusher --num-envs $n --no-cpu-baseline --param planar_spread=$sp 2>>$O/err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('Pusher n=$n spread=$sp %.4e ms/step %.4f'%(d['value'],d['ms_per_step']))" | tee -a $O/sweep.txt
done; done
