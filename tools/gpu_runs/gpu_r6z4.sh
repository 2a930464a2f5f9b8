#!/bin/bash
# Round 6, second session, closing pass: GPU suite + smoke + default bench, families table, bench lines of the MuJoCo families
set -u
export TMPDIR=/tmp
O=gpurun_out/r6z4; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q ) > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; grep -E "passed|failed|rc=|FAILED|real" $O/gpu_tests.log | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $O/gpu_tests.log 2>&1; tail -1 $O/gpu_tests.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-160 $O/bench_default.json
timeout 900 python tools/bench_families.py > $O/bench_families.md 2>>$O/err; tail -34 $O/bench_families.md | head -34
for cfg in "HalfCheetah 8192" "Walker2d 65536" "Hopper 65536" "Ant 32768" "Pusher 65536"; do
  set -- $cfg
  timeout 600 python bench.py --task $1 --num-envs $2 --no-cpu-baseline 2>>$O/err >> $O/bench.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r6z4/bench.jsonl'):
    d=json.loads(l); print(d['metric'], '%.3e'%d['value'], 'numpy %.3e' % d['numpy_api']['value'] if d.get('numpy_api') else '')
PY
