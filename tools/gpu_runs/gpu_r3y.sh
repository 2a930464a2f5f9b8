#!/bin/bash
# Humanoid / HumanoidStandup on the hybrid-PGS build: bench lines, then kernel trace + PMC passes (tools/profile_bench.sh)
set -u
export TMPDIR=/tmp
O=gpurun_out/r3y
mkdir -p $O
for t in HumanoidStandup Humanoid; do
  timeout 300 python bench.py --no-cpu-baseline --task $t --num-envs 65536 2>>$O/err >> $O/bench.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r3y/bench.jsonl'):
    d=json.loads(l); print(d['metric'], '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'])
PY
P() { tag=$1; shift; bash tools/profile_bench.sh $tag "$@" > $O/$tag.log 2>&1; }
P r3y_standup4 --task HumanoidStandup --num-envs 65536
P r3y_humanoid4 --task Humanoid --num-envs 65536
for d in gpurun_out/prof_r3y_*; do echo "== $d"; sed -n '/timed window/,/^$/p' $d/summary.md | head -4; grep -i "traffic\|FETCH\|WRITE" $d/summary.md | head -4; done
