#!/bin/bash
# HumanoidStandup with 24 register rows: profile (kernel trace + PMC passes) and bench lines
set -u
export TMPDIR=/tmp
O=gpurun_out/r2t
mkdir -p $O
bash tools/profile_bench.sh r2t_standup4 --task HumanoidStandup --num-envs 65536 > $O/p.log 2>&1
for cfg in "HumanoidStandup 65536" "HumanoidStandup 32768" "Humanoid 65536"; do
  set -- $cfg
  timeout 600 python bench.py --task $1 --num-envs $2 --no-cpu-baseline 2>>$O/err >> $O/bench.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r2t/bench.jsonl'):
    d=json.loads(l); print(d['metric'], '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'])
PY
grep -E "FETCH_SIZE|WRITE_SIZE|Humanoid4StepKernel<double>.*\| [0-9]+ \|" gpurun_out/prof_r2t_*/summary.md
