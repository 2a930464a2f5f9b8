#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2h
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mujoco.py -m gpu -x -q -s -k "pusher" > $O/gpu_tests_pusher.log 2>&1; echo "rc=$?" >> $O/gpu_tests_pusher.log; tail -8 $O/gpu_tests_pusher.log
for n in 16384 65536 262144; do
  timeout 300 python bench.py --task Pusher --num-envs $n --steps 100 --warmup 20 --no-cpu-baseline 2>>$O/err | tee -a $O/bench_pusher.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($n, '%.3e'%d['value'], 'kernel_ms', round(d['roofline']['kernel_ms'],3))"
done
tail -5 $O/err
