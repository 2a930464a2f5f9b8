#!/bin/bash
# Fourth measurement pass of round 3 (Humanoid TU changed again: lazy wave maximum): full GPU suite + smoke, default bench
# line, Humanoid / HumanoidStandup bench lines and kernel trace + PMC passes from one box.
set -u
export TMPDIR=/tmp
O=gpurun_out/r3zl
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; grep -E "passed|failed|rc=|FAILED" $O/gpu_tests.log | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $O/gpu_tests.log 2>&1; tail -1 $O/gpu_tests.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-200 $O/bench_default.json
for cfg in "Humanoid 65536" "HumanoidStandup 65536"; do
  set -- $cfg
  timeout 600 python bench.py --task $1 --num-envs $2 --no-cpu-baseline 2>>$O/err >> $O/bench.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r3zl/bench.jsonl'):
    d=json.loads(l); print(d['metric'], d['dtype'], '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'stale' if d['roofline'].get('stale') else '')
PY
P() { tag=$1; shift; bash tools/profile_bench.sh $tag "$@" > $O/$tag.log 2>&1; }
P r3zl_standup4 --task HumanoidStandup --num-envs 65536
P r3zl_humanoid4 --task Humanoid --num-envs 65536
