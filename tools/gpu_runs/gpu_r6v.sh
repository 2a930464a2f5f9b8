#!/bin/bash
# Round 6, second session: full GPU suite + smoke + default bench on the working tree
set -u
export TMPDIR=/tmp
O=gpurun_out/${TAG:-r6v}
mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q ) > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; grep -E "passed|failed|rc=|FAILED|real" $O/gpu_tests.log | tail -8
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $O/gpu_tests.log 2>&1; tail -1 $O/gpu_tests.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-600 $O/bench_default.json
