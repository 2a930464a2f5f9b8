#!/bin/bash
# Hybrid PGS of the Humanoid quad kernel (register rows + streamed overflow rows): Humanoid parity tests,
# then bench lines of HumanoidStandup / Humanoid for the product build and the A/B builds (register rows 20 / 16).
set -u
export TMPDIR=/tmp
O=gpurun_out/r3s
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_fullsize.py -m gpu -q -k "umanoid" > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; grep -E "passed|failed|rc=|FAILED" $O/gpu_tests.log | tail -6
B() { tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" 2>>$O/err | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); d['build']='$tag'; print(json.dumps(d))" >> $O/bench.jsonl; }
B product --task HumanoidStandup --num-envs 65536
B product --task Humanoid --num-envs 65536
cp envpool_amd/lib/libenvpool_amd.so /tmp/product.so
for tag in su20 su16; do
  [ -f envpool_amd/lib/libenvpool_amd_$tag.so ] || continue
  cp envpool_amd/lib/libenvpool_amd_$tag.so envpool_amd/lib/libenvpool_amd.so
  B $tag --task HumanoidStandup --num-envs 65536
  B $tag --task Humanoid --num-envs 65536
done
cp /tmp/product.so envpool_amd/lib/libenvpool_amd.so
python - <<'PY'
import json
for l in open('gpurun_out/r3s/bench.jsonl'):
    d=json.loads(l); print(d['build'], d['metric'], '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'])
PY
