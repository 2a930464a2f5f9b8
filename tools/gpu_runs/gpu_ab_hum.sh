#!/bin/bash
# same-box A/B of Humanoid-kernel libraries: product (envpool_amd/lib/libenvpool_amd.so as shipped in the snapshot)
# against envpool_amd/lib/libenvpool_amd_<tag>.so (tools/build_alt_hum4.sh), each twice, interleaved, both tasks.
#   usage: gpu_ab_hum.sh <out-name> <tag> [tag ...]
set -u
export TMPDIR=/tmp
O=gpurun_out/$1; shift
mkdir -p $O
cp envpool_amd/lib/libenvpool_amd.so /tmp/product.so
B() { timeout 300 python bench.py --no-cpu-baseline --task $2 --num-envs 65536 --steps 100 --min-time 0 2>>$O/err | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$1', d['metric'].split(', ')[-1], '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'])" | tee -a $O/ab.txt; }
for rep in 1 2; do
  cp /tmp/product.so envpool_amd/lib/libenvpool_amd.so; B product Humanoid; B product HumanoidStandup
  for v in "$@"; do cp envpool_amd/lib/libenvpool_amd_$v.so envpool_amd/lib/libenvpool_amd.so; B $v Humanoid; B $v HumanoidStandup; done
done
cp /tmp/product.so envpool_amd/lib/libenvpool_amd.so
