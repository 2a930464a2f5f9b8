#!/bin/bash
# Round 5, call aa: where 2 lanes per env overtake 4 on the final kernels (the pool picks 2 from 24576 rows up)
set -u
export TMPDIR=/tmp
O=gpurun_out/r5aa
mkdir -p $O
for n in 8192 12288 16384 20480 24576 32768 49152; do
for l in 2 4; do
  timeout 300 python bench.py --num-envs $n --no-cpu-baseline --min-time 1.5 --param planar_layout=$l 2>>$O/err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['num_envs_per_gpu'], 'layout', $l, '%.3e'%d['value'], 'kernel_ms %.4f'%d['roofline']['kernel_ms'])" | tee -a $O/layout_sweep.txt
done
done
