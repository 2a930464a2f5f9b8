#!/bin/bash
# round 3: full GPU suite (new config-sized parity tests, both Atari plugins) + smoke
set -u
export TMPDIR=/tmp
O=gpurun_out/r3b
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s --durations=15 > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; grep -E "passed|failed|rc=|Error|error" $O/gpu_tests.log | tail -8
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $O/gpu_tests.log 2>&1; tail -1 $O/gpu_tests.log
