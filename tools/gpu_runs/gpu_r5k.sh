#!/bin/bash
# Round 5, call k: line search of the lane-group solver cut to ONE evaluation (+ one Newton step of the 1-D problem from it)
# vs two vs the exact search (24 at most): planar parity tests on the variant, then interleaved bench lines
set -u
export TMPDIR=/tmp
O=gpurun_out/r5k
mkdir -p $O
cp envpool_amd/lib/libenvpool_amd.so /tmp/base.so
cp envpool_amd/lib/libenvpool_amd_ls1.so envpool_amd/lib/libenvpool_amd.so
( timeout 1200 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_mujoco_golden.py tests/test_gpu_fullsize.py -q ) > $O/gpu_tests_ls1.log 2>&1; grep -E "passed|failed|FAILED" $O/gpu_tests_ls1.log | tail -8
for rep in 1 2; do
for v in base ls1 ls2; do
  if [ $v = base ]; then cp /tmp/base.so envpool_amd/lib/libenvpool_amd.so; else cp envpool_amd/lib/libenvpool_amd_$v.so envpool_amd/lib/libenvpool_amd.so; fi
  for cfg in "HalfCheetah 65536" "HalfCheetah 8192" "Walker2d 65536" "Hopper 65536"; do
    set -- $cfg
    timeout 600 python bench.py --task $1 --num-envs $2 --no-cpu-baseline --min-time 2 2>>$O/err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['metric'].split(',')[-1], d['config']['num_envs_per_gpu'], '%.3e'%d['value'], 'kernel_ms %.4f'%d['roofline']['kernel_ms'], 'async %.3e'%d['async_mode']['value'])" | tee -a $O/ab.txt
  done
done
done
cp /tmp/base.so envpool_amd/lib/libenvpool_amd.so
