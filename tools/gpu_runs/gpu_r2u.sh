#!/bin/bash
# DPP moves with bound_ctrl (no init move of the destination): parity of every quad kernel + bench lines
set -u
export TMPDIR=/tmp
O=gpurun_out/r2u
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_fullsize.py tests/test_gpu_sharded.py -m gpu -x -q > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log
for cfg in "Humanoid 65536" "HumanoidStandup 65536" "Ant 65536" "Ant 32768" "Pusher 65536" "HalfCheetah 65536"; do
  set -- $cfg
  timeout 600 python bench.py --task $1 --num-envs $2 --no-cpu-baseline 2>>$O/err >> $O/bench.jsonl
done
tail -3 $O/tests.log
python - <<'PY'
import json
for l in open('gpurun_out/r2u/bench.jsonl'):
    d=json.loads(l); print(d['metric'], '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'])
PY
