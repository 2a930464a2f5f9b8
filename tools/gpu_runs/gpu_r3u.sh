#!/bin/bash
# Hybrid PGS with on-chip overflow rows: A/B over (register rows, overflow rows) of HumanoidStandup;
# Humanoid parity tests on the first build.
set -u
export TMPDIR=/tmp
O=gpurun_out/r3u
mkdir -p $O
cp envpool_amd/lib/libenvpool_amd.so /tmp/product.so
B() { tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" 2>>$O/err | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); d['build']='$tag'; print(json.dumps(d))" >> $O/bench.jsonl; }
first=1
for tag in s16c16 s20c8 s20c12 s16c12 s24c8; do
  [ -f envpool_amd/lib/libenvpool_amd_$tag.so ] || continue
  cp envpool_amd/lib/libenvpool_amd_$tag.so envpool_amd/lib/libenvpool_amd.so
  if [ $first = 1 ]; then
    timeout 900 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_fullsize.py -m gpu -q -k "umanoid" > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; grep -E "passed|failed|rc=|FAILED" $O/gpu_tests.log | tail -6
    B $tag --task Humanoid --num-envs 65536
    first=0
  fi
  B $tag --task HumanoidStandup --num-envs 65536 --steps 100 --min-time 0
done
cp /tmp/product.so envpool_amd/lib/libenvpool_amd.so
python - <<'PY'
import json
for l in open('gpurun_out/r3u/bench.jsonl'):
    d=json.loads(l); print(d['build'], d['metric'], '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'])
PY
