#!/bin/bash
# headline kernel: throughput vs num_envs on the final build (window timing)
set -u
export TMPDIR=/tmp
O=gpurun_out/r2zd
mkdir -p $O
for n in 8192 16384 32768 65536 131072 262144 1048576; do
  timeout 600 python bench.py --num-envs $n --no-cpu-baseline 2>>$O/err >> $O/bench_sweep.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r2zd/bench_sweep.jsonl'):
    d=json.loads(l); print(d['config']['num_envs_per_gpu'], '%.4e'%d['value'], 'ms/step %.4f'%d['ms_per_step'])
PY
