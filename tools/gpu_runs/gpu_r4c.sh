#!/bin/bash
# round 4, call c: GPU suite after the fp32 removal, the bench line (cpu_baseline: threadpool + OpenMP),
# then the lane-group stage timers (diagnostic library swapped in last)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r4c_tests.log
timeout 600 python bench.py > gpurun_out/r4c_bench.json 2> gpurun_out/r4c_bench.err
cp envpool_amd/lib/libenvpool_amd_lgtimers.so envpool_amd/lib/libenvpool_amd.so
( timeout 300 python tools/lg_stage_timers.py HalfCheetah 65536 100
  timeout 300 python tools/lg_stage_timers.py HalfCheetah 8192 100
  timeout 300 python tools/lg_stage_timers.py Walker2d 65536 50 ) > gpurun_out/r4c_lg_timers.log 2>&1
cat gpurun_out/r4c_tests.log gpurun_out/r4c_lg_timers.log; tail -c 1200 gpurun_out/r4c_bench.json
