#!/bin/bash
# round 3: Pusher with the constraint-row Jacobians in LDS (touching rows only): parity, bench, profile
set -u
export TMPDIR=/tmp
O=gpurun_out/r3n
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "usher" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=|FAILED" $O/tests.log | tail -5
timeout 300 python tools/hum_poison_check.py Pusher > $O/poison.log 2>&1; cat $O/poison.log
for n in 16384 32768 65536 131072; do
  timeout 300 python bench.py --task Pusher --num-envs $n --no-cpu-baseline 2>>$O/err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('Pusher n=$n %.4e ms/step %.4f'%(d['value'],d['ms_per_step']))" | tee -a $O/pusher.txt
done
bash tools/profile_bench.sh r3n_pusher --task Pusher --num-envs 65536 > /dev/null 2>&1; sed -n '/timed window/,$p' gpurun_out/prof_r3n_pusher/summary.md
