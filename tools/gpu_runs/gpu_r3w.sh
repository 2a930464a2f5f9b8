#!/bin/bash
# stage timers of the (16, 16) hybrid + A/B of the DPP bound_ctrl form on the leaner visits
set -u
export TMPDIR=/tmp
O=gpurun_out/r3w
mkdir -p $O
cp envpool_amd/lib/libenvpool_amd.so /tmp/product.so
cp envpool_amd/lib/libenvpool_amd_s16c16dbg.so envpool_amd/lib/libenvpool_amd.so
timeout 600 python tools/hum_solver_stats.py HumanoidStandup 16384 > $O/standup_stats.txt 2>&1; tail -9 $O/standup_stats.txt
for tag in s16c16bc; do
  cp envpool_amd/lib/libenvpool_amd_$tag.so envpool_amd/lib/libenvpool_amd.so
  for t in HumanoidStandup Humanoid; do
  timeout 300 python bench.py --no-cpu-baseline --task $t --num-envs 65536 --steps 100 --min-time 0 2>>$O/err | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$tag', d['metric'], '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'])"
  done
done
cp /tmp/product.so envpool_amd/lib/libenvpool_amd.so
