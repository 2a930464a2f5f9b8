#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2q
mkdir -p $O
bash tools/profile_bench.sh r2q_humanoid4 --task Humanoid --num-envs 65536 > $O/p5.log 2>&1
bash tools/profile_bench.sh r2q_standup4 --task HumanoidStandup --num-envs 65536 > $O/p6.log 2>&1
for cfg in "Humanoid 65536" "HumanoidStandup 65536" "Humanoid 32768" "Humanoid 131072"; do
  set -- $cfg
  timeout 600 python bench.py --task $1 --num-envs $2 --no-cpu-baseline 2>>$O/err >> $O/bench.jsonl
done
timeout 900 python bench.py --task Humanoid --num-envs 65536 2>>$O/err > $O/bench_humanoid.json
python - <<'PY'
import json
for l in open('gpurun_out/r2q/bench.jsonl'):
    d=json.loads(l); print(d['metric'], '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'])
d=json.load(open('gpurun_out/r2q/bench_humanoid.json')); print('cpu_baseline', d['cpu_baseline'])
PY
grep -E "FETCH_SIZE|WRITE_SIZE|Humanoid4StepKernel<double>.*\| [0-9]+ \|" gpurun_out/prof_r2q_*/summary.md
