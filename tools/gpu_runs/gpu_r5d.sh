#!/bin/bash
# Round 5, call d: tile-wise lazy mt19937 regeneration -- bit-exact suites, the HBM-streaming families again,
# Hopper / Walker2d / HalfCheetah bench lines
set -u
export TMPDIR=/tmp
O=gpurun_out/r5d
mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q -x ) > $O/gpu_tests.log 2>&1; grep -E "passed|failed|real" $O/gpu_tests.log | tail -4
bash tools/profile_families.sh r5d_families --families CartPole,MountainCar,Pendulum,Acrobot,FrozenLake,NChain,Blackjack,Catch,Taxi,CliffWalking > $O/families.log 2>&1
tail -26 $O/families.log
for cfg in "Hopper 65536" "Walker2d 65536" "HalfCheetah 65536" "Ant 32768" "Humanoid 65536"; do
  set -- $cfg
  timeout 600 python bench.py --task $1 --num-envs $2 --no-cpu-baseline --min-time 2 2>>$O/err >> $O/bench.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r5d/bench.jsonl'):
    d=json.loads(l); print(d['metric'], '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'async %.3e'%d['async_mode']['value'], 'reset_ms %.3f'%d['reset_step_ms'])
PY
