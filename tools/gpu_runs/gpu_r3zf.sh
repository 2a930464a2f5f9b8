#!/bin/bash
# same-box A/B of the row loop's register cache (trunk cdofs / trunk block of the factor) for HumanoidStandup
set -u
export TMPDIR=/tmp
O=gpurun_out/r3zf
mkdir -p $O
cp envpool_amd/lib/libenvpool_amd.so /tmp/product.so
B() { timeout 300 python bench.py --no-cpu-baseline --task $2 --num-envs 65536 --steps 100 --min-time 0 2>>$O/err | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$1', d['metric'].split(', ')[-1], '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'])" | tee -a $O/ab.txt; }
for rep in 1 2; do
  cp /tmp/product.so envpool_amd/lib/libenvpool_amd.so; B product HumanoidStandup
  for v in rc1 rc2 rc3; do cp envpool_amd/lib/libenvpool_amd_$v.so envpool_amd/lib/libenvpool_amd.so; B $v HumanoidStandup; done
done
cp /tmp/product.so envpool_amd/lib/libenvpool_amd.so
