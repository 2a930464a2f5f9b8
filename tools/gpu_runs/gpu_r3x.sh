#!/bin/bash
# Product build after the Humanoid / HumanoidStandup solver work: Humanoid-family GPU tests and both bench lines
set -u
export TMPDIR=/tmp
O=gpurun_out/r3x
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_fullsize.py tests/test_gpu_device_path.py -m gpu -q > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; grep -E "passed|failed|rc=|FAILED" $O/gpu_tests.log | tail -6
for t in HumanoidStandup Humanoid; do
  timeout 300 python bench.py --no-cpu-baseline --task $t --num-envs 65536 2>>$O/err >> $O/bench.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r3x/bench.jsonl'):
    d=json.loads(l); print(d['metric'], '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'])
PY
