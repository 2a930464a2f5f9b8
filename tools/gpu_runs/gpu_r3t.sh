#!/bin/bash
# HumanoidStandup with the hybrid PGS (A/B build: 20 register rows, diagnostic switches compiled in):
# solver statistics per env-step and the stage breakdown (stages switched off one by one).
set -u
export TMPDIR=/tmp
O=gpurun_out/r3t
mkdir -p $O
cp envpool_amd/lib/libenvpool_amd.so /tmp/product.so
cp envpool_amd/lib/libenvpool_amd_su20dbg.so envpool_amd/lib/libenvpool_amd.so
timeout 300 python tools/hum_solver_stats.py HumanoidStandup 16384 > $O/standup_stats.txt 2>&1; cat $O/standup_stats.txt | tail -8
for dbg in 0 8 1 2 6; do
  timeout 300 python bench.py --no-cpu-baseline --task HumanoidStandup --num-envs 65536 --steps 60 --warmup 10 --min-time 0 --param hum_debug=$dbg 2>>$O/err >> $O/stages.jsonl
done
cp /tmp/product.so envpool_amd/lib/libenvpool_amd.so
python - <<'PY'
import json
for l in open('gpurun_out/r3t/stages.jsonl'):
    d=json.loads(l); print(d['config']['params'], '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'])
PY
