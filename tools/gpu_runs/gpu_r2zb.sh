#!/bin/bash
# bench lines with actions drawn in each task's action space (Humanoid / Standup +-0.4, Pusher +-2)
# next to the +-1 draw used until now
set -u
export TMPDIR=/tmp
O=gpurun_out/r2zb
mkdir -p $O
for cfg in "Humanoid 65536" "HumanoidStandup 65536" "Pusher 65536"; do
  set -- $cfg
  timeout 600 python bench.py --task $1 --num-envs $2 --no-cpu-baseline 2>>$O/err >> $O/bench.jsonl
  timeout 600 python bench.py --task $1 --num-envs $2 --no-cpu-baseline --action-scale 1.0 2>>$O/err >> $O/bench.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r2zb/bench.jsonl'):
    d=json.loads(l); print(d['metric'], '%.4e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], d['config']['workload'][40:110])
PY
