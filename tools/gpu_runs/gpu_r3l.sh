#!/bin/bash
# round 3: partly filled waves for small batches (lane-group kernel): parity + A/B over batch sizes
set -u
export TMPDIR=/tmp
O=gpurun_out/r3l
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_api.py tests/test_gpu_fullsize.py -m gpu -q -x -k "lane_group or spread or teacher_forced_step or walker or async or headline or config3 or frame_stack or batch_composition" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=|FAILED" $O/tests.log | tail -8
for n in 2048 4096 8192 12288 16384 24576 28672 32768; do for sp in 1 0; do
  timeout 300 python bench.py --num-envs $n --no-cpu-baseline --min-time 0.5 --param planar_spread=$sp 2>>$O/err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('HalfCheetah n=$n spread=$sp %.4e ms/step %.4f'%(d['value'],d['ms_per_step']))" | tee -a $O/spread_ab.txt
done; done
for n in 8192 16384; do for sp in 1 0; do
  timeout 300 python bench.py --task Walker2d --num-envs $n --no-cpu-baseline --min-time 0.5 --param planar_spread=$sp 2>>$O/err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('Walker2d n=$n spread=$sp %.4e ms/step %.4f'%(d['value'],d['ms_per_step']))" | tee -a $O/spread_ab.txt
done; done
