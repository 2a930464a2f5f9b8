#!/bin/bash
# same-box A/B for Humanoid: row-loop register cache (factor only / cdofs only), 16 register rows
set -u
export TMPDIR=/tmp
O=gpurun_out/r3zg
mkdir -p $O
cp envpool_amd/lib/libenvpool_amd.so /tmp/product.so
B() { timeout 300 python bench.py --no-cpu-baseline --task $2 --num-envs 65536 --steps 100 --min-time 0 2>>$O/err | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$1', d['metric'].split(', ')[-1], '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'])" | tee -a $O/ab.txt; }
for rep in 1 2; do
  cp /tmp/product.so envpool_amd/lib/libenvpool_amd.so; B product Humanoid
  for v in hrc2 hrc3 hr16; do cp envpool_amd/lib/libenvpool_amd_$v.so envpool_amd/lib/libenvpool_amd.so; B $v Humanoid; done
done
cp /tmp/product.so envpool_amd/lib/libenvpool_amd.so
