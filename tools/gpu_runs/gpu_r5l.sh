#!/bin/bash
# Round 5, call l: the one-evaluation line search as the product build -- full GPU suite, soak of the trip counts,
# bench lines
set -u
export TMPDIR=/tmp
O=gpurun_out/r5l
mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q ) > $O/gpu_tests.log 2>&1; grep -E "passed|failed|FAILED|real" $O/gpu_tests.log | tail -8
timeout 900 python tools/lg_iter_soak.py 400 > $O/iter_soak.txt 2>>$O/err; cat $O/iter_soak.txt | cut -c1-400
for cfg in "HalfCheetah 65536" "HalfCheetah 8192" "Walker2d 65536" "Hopper 65536" "Ant 32768" "Ant 65536"; do
  set -- $cfg
  timeout 600 python bench.py --task $1 --num-envs $2 --no-cpu-baseline --min-time 2 2>>$O/err >> $O/bench.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r5l/bench.jsonl'):
    d=json.loads(l); print(d['metric'].split(',')[-1], d['config']['num_envs_per_gpu'], '%.3e'%d['value'], 'kernel_ms %.4f'%d['roofline']['kernel_ms'], 'async %.3e'%d['async_mode']['value'])
PY
