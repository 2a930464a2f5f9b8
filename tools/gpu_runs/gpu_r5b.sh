#!/bin/bash
# Round 5, call b: the chunk schedule inside one launch of the headline kernel (diagnostic build -DEPA_LG_SCHED_TRACE,
# swapped in on the box only)
set -u
export TMPDIR=/tmp
O=gpurun_out/r5b
mkdir -p $O
cp envpool_amd/lib/libenvpool_amd_sched.so envpool_amd/lib/libenvpool_amd.so
for cfg in "HalfCheetah 65536" "HalfCheetah 32768" "HalfCheetah 131072" "Walker2d 65536" "Hopper 65536"; do
  set -- $cfg
  timeout 300 python tools/lg_sched_trace.py $1 $2 20 >> $O/lg_sched_trace.txt 2>> $O/err
done
timeout 300 python tools/lg_sched_trace.py HalfCheetah 65536 20 planar_lpt=0 >> $O/lg_sched_trace.txt 2>> $O/err
cat $O/lg_sched_trace.txt; tail -3 $O/err
