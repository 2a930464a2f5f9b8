#!/bin/bash
# Standup with 24 register rows: parity + bench
set -u
export TMPDIR=/tmp
O=gpurun_out/r2s
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mujoco.py -m gpu -x -q -k "umanoid" > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log
for cfg in "HumanoidStandup 65536" "Humanoid 65536"; do
  set -- $cfg
  timeout 600 python bench.py --task $1 --num-envs $2 --no-cpu-baseline 2>>$O/err >> $O/bench.jsonl
done
tail -3 $O/tests.log
python - <<'PY'
import json
for l in open('gpurun_out/r2s/bench.jsonl'):
    d=json.loads(l); print(d['metric'], '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'])
PY
