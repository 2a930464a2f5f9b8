#!/bin/bash
# round 3: full GPU suite on the clean rebuild (Pusher LDS rows, Ant / Pusher line-search termination) and
# async mode with fewer, larger batches in flight
set -u
export TMPDIR=/tmp
O=gpurun_out/r3o
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; grep -E "passed|failed|rc=|FAILED" $O/gpu_tests.log | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $O/gpu_tests.log 2>&1; tail -1 $O/gpu_tests.log
timeout 900 python tools/bench_async_api.py streams 2>>$O/err | tee $O/async_streams.jsonl
