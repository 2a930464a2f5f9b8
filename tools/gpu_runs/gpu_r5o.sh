#!/bin/bash
# Round 5, call o: stage timers of the lane-group kernel after the one-evaluation line search (diagnostic library)
set -u
export TMPDIR=/tmp
O=gpurun_out/r5o
mkdir -p $O
cp envpool_amd/lib/libenvpool_amd.so /tmp/base.so
cp envpool_amd/lib/libenvpool_amd_lgtimers.so envpool_amd/lib/libenvpool_amd.so
( timeout 300 python tools/lg_stage_timers.py HalfCheetah 65536 100
  timeout 300 python tools/lg_stage_timers.py HalfCheetah 8192 100
  timeout 300 python tools/lg_stage_timers.py HalfCheetah 32768 100
  timeout 300 python tools/lg_stage_timers.py HalfCheetah 8192 100 2
  timeout 300 python tools/lg_stage_timers.py Walker2d 65536 50
  timeout 300 python tools/lg_stage_timers.py Hopper 65536 50 ) > $O/lg_stage_timers.txt 2>&1
cp /tmp/base.so envpool_amd/lib/libenvpool_amd.so
cat $O/lg_stage_timers.txt
