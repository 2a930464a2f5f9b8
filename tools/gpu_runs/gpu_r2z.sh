#!/bin/bash
# window timing in bench.py (no event between launches): default line x2, per-family lines, contract + device-path tests
set -u
export TMPDIR=/tmp
O=gpurun_out/r2z
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_device_path.py tests/test_gpu_api.py -m gpu -q > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log
tail -3 $O/tests.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --no-cpu-baseline >> $O/bench.jsonl 2>>$O/err
for cfg in "Walker2d 65536" "Hopper 65536" "Ant 32768" "Ant 65536" "Pusher 65536" "Humanoid 65536" "HumanoidStandup 65536"; do
  set -- $cfg
  timeout 600 python bench.py --task $1 --num-envs $2 --no-cpu-baseline 2>>$O/err >> $O/bench.jsonl
done
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2z/bench_default.json')); print('default', '%.4e'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'kernel_ms %.4f'%d['roofline']['kernel_ms'], 'frac %.3f'%d['roofline']['frac'], d['cpu_baseline']['value'])
for l in open('gpurun_out/r2z/bench.jsonl'):
    d=json.loads(l); print(d['metric'], '%.4e'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'kernel_ms %.4f'%d['roofline']['kernel_ms'])
PY
