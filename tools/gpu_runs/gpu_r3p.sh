#!/bin/bash
# round 3: bench.py with the async-mode leg
set -u
export TMPDIR=/tmp
O=gpurun_out/r3p
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bench_contract.py -m gpu -q > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.load(open('$O/bench_default.json')); print('value %.4e'%d['value'], d['ms_per_step'], 'async', d['async_mode'], 'numpy', d['numpy_api']['value'], 'roof', d['roofline']['frac'], d['roofline'].get('stale'), d['roofline']['traffic'])"
for t in Ant Humanoid Pusher Walker2d; do timeout 600 python bench.py --task $t --no-cpu-baseline 2>>$O/err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$t value %.4e'%d['value'], 'async %.4e'%d['async_mode']['value'], 'stale', d['roofline'].get('stale'))"; done
