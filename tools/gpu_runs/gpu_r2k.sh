#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2k
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_api.py -m gpu -x -q -k "humanoid or Humanoid or composition" > $O/gpu_tests_hum.log 2>&1; echo "rc=$?" >> $O/gpu_tests_hum.log; tail -3 $O/gpu_tests_hum.log
for task in HumanoidStandup Humanoid; do
  timeout 300 python bench.py --task $task --num-envs 65536 --steps 100 --warmup 20 --no-cpu-baseline 2>>$O/err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$task', '%.3e'%d['value'], 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'ms/step', round(d['ms_per_step'],3))" | tee -a $O/stages.txt
done
bash tools/gpu_runs/gpu_r2l.sh | tail -3
