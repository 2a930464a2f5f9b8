set -u
export TMPDIR=/tmp
O=gpurun_out/quick
mkdir -p $O; rm -f $O/bench.jsonl
( timeout 1200 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_mujoco_golden.py tests/test_gpu_fullsize.py -q ) > $O/gpu_tests.log 2>&1; grep -E "passed|failed|FAILED" $O/gpu_tests.log | tail -5
for cfg in "HalfCheetah 65536" "HalfCheetah 8192" "Walker2d 65536" "Hopper 65536"; do
  set -- $cfg
  timeout 600 python bench.py --task $1 --num-envs $2 --no-cpu-baseline --min-time 2 2>>$O/err >> $O/bench.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/quick/bench.jsonl'):
    d=json.loads(l); print(d['metric'].split(',')[-1], d['config']['num_envs_per_gpu'], '%.3e'%d['value'], 'kernel_ms %.4f'%d['roofline']['kernel_ms'], 'async %.3e'%d['async_mode']['value'])
PY
