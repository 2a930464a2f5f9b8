#!/bin/bash
# same-box A/B of the call around the constraint stage: product (Humanoid inlined, HumanoidStandup behind the call),
# `hcall` (Humanoid behind the call too), `sinl` (HumanoidStandup inlined); each library twice, interleaved
set -u
export TMPDIR=/tmp
O=gpurun_out/r3ze
mkdir -p $O
cp envpool_amd/lib/libenvpool_amd.so /tmp/product.so
B() { timeout 300 python bench.py --no-cpu-baseline --task $2 --num-envs 65536 --steps 100 --min-time 0 2>>$O/err | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$1', d['metric'].split(', ')[-1], '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'])" | tee -a $O/ab.txt; }
for rep in 1 2; do
  cp /tmp/product.so envpool_amd/lib/libenvpool_amd.so; B product Humanoid; B product HumanoidStandup
  cp envpool_amd/lib/libenvpool_amd_hcall.so envpool_amd/lib/libenvpool_amd.so; B hcall Humanoid
  cp envpool_amd/lib/libenvpool_amd_sinl.so envpool_amd/lib/libenvpool_amd.so; B sinl HumanoidStandup
done
cp /tmp/product.so envpool_amd/lib/libenvpool_amd.so
