#!/bin/bash
# Round 5, call j: BASELINE config 3 (HalfCheetah N=8192) under the layout / spread knobs that exist
set -u
export TMPDIR=/tmp
O=gpurun_out/r5j
mkdir -p $O
for p in "" "--param planar_layout=2" "--param planar_layout=2 --param planar_spread=0" "--param planar_layout=4 --param planar_spread=0" "--param planar_layout=4 --param planar_waves=2" "--param planar_lpt=0"; do
  timeout 300 python bench.py --num-envs 8192 --no-cpu-baseline --min-time 2 $p 2>>$O/err >> $O/bench.jsonl
done
for n in 16384 24576 32768 49152; do
  timeout 300 python bench.py --num-envs $n --no-cpu-baseline --min-time 2 2>>$O/err >> $O/bench.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r5j/bench.jsonl'):
    d=json.loads(l); print(d['config']['num_envs_per_gpu'], d['config']['params'], '%.3e'%d['value'], 'kernel_ms %.4f'%d['roofline']['kernel_ms'], 'async %.3e'%d['async_mode']['value'])
PY
