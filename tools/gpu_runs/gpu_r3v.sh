#!/bin/bash
# Stage timers + solver statistics of HumanoidStandup on a diagnostic A/B build ($1 = tag of the library)
set -u
export TMPDIR=/tmp
TAG=${1:-s20c8dbg}
O=gpurun_out/r3v
mkdir -p $O
cp envpool_amd/lib/libenvpool_amd.so /tmp/product.so
cp envpool_amd/lib/libenvpool_amd_$TAG.so envpool_amd/lib/libenvpool_amd.so
timeout 600 python tools/hum_solver_stats.py ${2:-HumanoidStandup} 16384 > $O/standup_stats_$TAG.txt 2>&1; tail -16 $O/standup_stats_$TAG.txt
cp /tmp/product.so envpool_amd/lib/libenvpool_amd.so
