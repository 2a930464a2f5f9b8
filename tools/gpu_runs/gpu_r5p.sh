#!/bin/bash
# Round 5, call p: broad phase in front of the Hopper's capsule-capsule narrow phases -- planar parity tests + bench lines
set -u
export TMPDIR=/tmp
O=gpurun_out/r5p
mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_mujoco_golden.py tests/test_gpu_fullsize.py -q ) > $O/gpu_tests.log 2>&1; grep -E "passed|failed|FAILED" $O/gpu_tests.log | tail -5
for cfg in "Hopper 65536" "Hopper 131072" "HalfCheetah 65536"; do
  set -- $cfg
  timeout 600 python bench.py --task $1 --num-envs $2 --no-cpu-baseline --min-time 2 2>>$O/err >> $O/bench.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r5p/bench.jsonl'):
    d=json.loads(l); print(d['metric'].split(',')[-1], d['config']['num_envs_per_gpu'], '%.3e'%d['value'], 'kernel_ms %.4f'%d['roofline']['kernel_ms'], 'async %.3e'%d['async_mode']['value'])
PY
