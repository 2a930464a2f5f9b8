#!/bin/bash
# Round 5, call n: the batch's completion event recorded only when a consumer needs it (single-stream pools) --
# full GPU suite, then the launch-latency-bound families and the headline, against the round-5 pass (r5z, an event per launch)
set -u
export TMPDIR=/tmp
O=gpurun_out/r5n
mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q ) > $O/gpu_tests.log 2>&1; grep -E "passed|failed|FAILED|real" $O/gpu_tests.log | tail -8
timeout 900 python tools/bench_families.py --no-atari --big 0 2>>$O/err | grep "^|" > $O/bench_families_65536.md; cat $O/bench_families_65536.md
for cfg in "HalfCheetah 65536" "HalfCheetah 8192" "Hopper 65536"; do
  set -- $cfg
  timeout 600 python bench.py --task $1 --num-envs $2 --no-cpu-baseline --min-time 2 2>>$O/err >> $O/bench.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r5n/bench.jsonl'):
    d=json.loads(l); print(d['metric'].split(',')[-1], d['config']['num_envs_per_gpu'], '%.3e'%d['value'], 'ms_per_step %.4f'%d['ms_per_step'], 'kernel_ms %.4f'%d['roofline']['kernel_ms'], 'async %.3e'%d['async_mode']['value'], 'numpy %.3e'%d['numpy_api']['value'])
PY
