#!/bin/bash
# round 4, call b: the whole GPU suite on the round-4 tree, the default bench line, then the stage timers of
# the lane-group kernel (diagnostic library swapped in LAST: the snapshot on the GPU box is disposable)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r4b_tests.log
timeout 600 python bench.py > gpurun_out/r4b_bench.json 2> gpurun_out/r4b_bench.err
cp envpool_amd/lib/libenvpool_amd_lgtimers.so envpool_amd/lib/libenvpool_amd.so
( timeout 300 python tools/lg_stage_timers.py HalfCheetah 65536 100
  timeout 300 python tools/lg_stage_timers.py HalfCheetah 8192 100
  timeout 300 python tools/lg_stage_timers.py Walker2d 65536 50 ) > gpurun_out/r4b_lg_timers.log 2>&1
cat gpurun_out/r4b_tests.log gpurun_out/r4b_lg_timers.log; tail -c 1500 gpurun_out/r4b_bench.json
