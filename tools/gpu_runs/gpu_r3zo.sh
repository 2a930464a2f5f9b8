#!/bin/bash
# Last pass of round 3 for the Humanoid TU (stage timers routed for Humanoid too; comments): Humanoid-family GPU tests,
# both bench lines, kernel trace + PMC passes (sources of the pmc.json entries)
set -u
export TMPDIR=/tmp
O=gpurun_out/r3zo
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_fullsize.py tests/test_gpu_sharded.py -m gpu -q > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; grep -E "passed|failed|rc=" $O/gpu_tests.log | tail -3
for cfg in "Humanoid 65536" "HumanoidStandup 65536"; do
  set -- $cfg
  timeout 600 python bench.py --task $1 --num-envs $2 --no-cpu-baseline 2>>$O/err >> $O/bench.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r3zo/bench.jsonl'):
    d=json.loads(l); print(d['metric'], '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'])
PY
P() { tag=$1; shift; bash tools/profile_bench.sh $tag "$@" > $O/$tag.log 2>&1; }
P r3zo_standup4 --task HumanoidStandup --num-envs 65536
P r3zo_humanoid4 --task Humanoid --num-envs 65536
