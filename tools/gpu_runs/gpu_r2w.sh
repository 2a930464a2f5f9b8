#!/bin/bash
# numpy API (PCIe inclusive) for the round-2 kernels, and kernel rate vs num_envs for Ant / Humanoid / Pusher
set -u
export TMPDIR=/tmp
O=gpurun_out/r2w
mkdir -p $O
for cfg in "HalfCheetah-v4 65536 100" "Ant-v4 32768 60" "Ant-v4 65536 60" "Humanoid-v4 65536 30" "HumanoidStandup-v4 65536 12" "Pusher-v4 65536 100" "Walker2d-v4 65536 100"; do
  set -- $cfg
  timeout 300 python tools/bench_numpy_api.py $1 $2 $3 >> $O/numpy_api.jsonl 2>>$O/err
done
for cfg in "Ant 16384" "Ant 131072" "Pusher 32768" "Pusher 131072" "Humanoid 16384" "HumanoidStandup 131072"; do
  set -- $cfg
  timeout 600 python bench.py --task $1 --num-envs $2 --no-cpu-baseline 2>>$O/err >> $O/bench_sweep.jsonl
done
cat $O/numpy_api.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r2w/bench_sweep.jsonl'):
    d=json.loads(l); print(d['metric'], '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'])
PY
