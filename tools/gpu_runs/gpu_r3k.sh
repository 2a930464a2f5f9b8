#!/bin/bash
# round 3: fixes after the mid-round suite (frame stack through the generic ring, concurrent launches only
# for families without per-launch scratch) + compile-flag A/B of the lane-group TU
set -u
export TMPDIR=/tmp
O=gpurun_out/r3k
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; grep -E "passed|failed|rc=|FAILED" $O/gpu_tests.log | tail -8
L=envpool_amd/lib
cp $L/libenvpool_amd.so $L/libenvpool_amd_main.so
for tag in main licm ilp main; do
  cp $L/libenvpool_amd_$tag.so $L/libenvpool_amd.so
  for n in 65536 8192; do
    timeout 300 python bench.py --num-envs $n --no-cpu-baseline --min-time 1 2>>$O/err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag HalfCheetah n=$n %.4e ms/step %.4f'%(d['value'],d['ms_per_step']))" | tee -a $O/flags_ab.txt
  done
done
cp $L/libenvpool_amd_main.so $L/libenvpool_amd.so
