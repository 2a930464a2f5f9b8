#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2j
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_api.py -m gpu -x -q -s -k "humanoid or Humanoid or composition" > $O/gpu_tests_hum.log 2>&1; echo "rc=$?" >> $O/gpu_tests_hum.log; tail -12 $O/gpu_tests_hum.log
for lay in 1 0; do for task in Humanoid HumanoidStandup; do
  timeout 300 python bench.py --task $task --num-envs 65536 --steps 20 --warmup 5 --no-cpu-baseline --param hum_layout=$lay 2>>$O/err | tee -a $O/bench_hum.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$task layout $lay', '%.3e'%d['value'], 'kernel_ms', round(d['roofline']['kernel_ms'],3))"
done; done
tail -3 $O/err
