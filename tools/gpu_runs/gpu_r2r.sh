#!/bin/bash
# end-of-round check: full GPU suite, default bench line, Humanoid bench line, per-family table
set -u
export TMPDIR=/tmp
O=gpurun_out/r2r
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -s > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; grep -E "passed|failed|rc=" $O/gpu_tests.log | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $O/gpu_tests.log 2>&1; tail -1 $O/gpu_tests.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
timeout 900 python bench.py --task Humanoid --num-envs 65536 2>>$O/err > $O/bench_humanoid.json; cut -c1-200 $O/bench_humanoid.json
timeout 900 python tools/bench_families.py --steps 100 > $O/bench_families.md 2>>$O/err; tail -32 $O/bench_families.md
