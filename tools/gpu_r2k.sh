#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2k
mkdir -p $O
for task in Humanoid HumanoidStandup; do
  timeout 300 python bench.py --task $task --num-envs 65536 --steps 30 --warmup 10 --no-cpu-baseline 2>>$O/err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$task', '%.3e'%d['value'], 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'ms/step', round(d['ms_per_step'],3))" | tee -a $O/stages.txt
done
timeout 300 python bench.py --task Humanoid --num-envs 65536 --steps 200 --warmup 20 --no-cpu-baseline 2>>$O/err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('Humanoid 200 steps', '%.3e'%d['value'], 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'ms/step', round(d['ms_per_step'],3))" | tee -a $O/stages.txt
