#!/bin/bash
# Round 5, call w: chunk schedule of the lane-group kernel on the final build (diagnostic -DEPA_LG_SCHED_TRACE library)
set -u
export TMPDIR=/tmp
O=gpurun_out/r5w
mkdir -p $O
cp envpool_amd/lib/libenvpool_amd.so /tmp/new.so
cp envpool_amd/lib/libenvpool_amd_sched.so envpool_amd/lib/libenvpool_amd.so
for cfg in "HalfCheetah 65536" "HalfCheetah 131072" "Walker2d 65536" "Hopper 65536"; do
  set -- $cfg
  timeout 300 python tools/lg_sched_trace.py $1 $2 20 >> $O/lg_sched_trace.txt 2>> $O/err
done
cp /tmp/new.so envpool_amd/lib/libenvpool_amd.so
grep -E "N=|span|busy|  mean|  std|  mx |ideal|lpt|n1|n2|n3" $O/lg_sched_trace.txt
