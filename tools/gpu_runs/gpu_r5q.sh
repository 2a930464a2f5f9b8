#!/bin/bash
# Round 5, call q: a lane visits its OWN touching slots (per-lane slot sets) instead of the wave's union -- planar parity
# tests on the new build, then interleaved bench lines against the previous build (libenvpool_amd_prev.so)
set -u
export TMPDIR=/tmp
O=gpurun_out/r5q
mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_mujoco_golden.py tests/test_gpu_fullsize.py -q ) > $O/gpu_tests.log 2>&1; grep -E "passed|failed|FAILED" $O/gpu_tests.log | tail -8
cp envpool_amd/lib/libenvpool_amd.so /tmp/new.so
for rep in 1 2; do
for v in prev new; do
  if [ $v = new ]; then cp /tmp/new.so envpool_amd/lib/libenvpool_amd.so; else cp envpool_amd/lib/libenvpool_amd_prev.so envpool_amd/lib/libenvpool_amd.so; fi
  for cfg in "HalfCheetah 65536" "HalfCheetah 8192" "Walker2d 65536" "Hopper 65536"; do
    set -- $cfg
    timeout 600 python bench.py --task $1 --num-envs $2 --no-cpu-baseline --min-time 2 2>>$O/err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['metric'].split(',')[-1], d['config']['num_envs_per_gpu'], '%.3e'%d['value'], 'kernel_ms %.4f'%d['roofline']['kernel_ms'], 'async %.3e'%d['async_mode']['value'])" | tee -a $O/ab.txt
  done
done
done
cp /tmp/new.so envpool_amd/lib/libenvpool_amd.so
