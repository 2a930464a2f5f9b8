#!/bin/bash
# planar kernel A/B harness: parity of the planar families + bench lines (two repetitions)
set -u
export TMPDIR=/tmp
O=gpurun_out/r2y
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_fullsize.py -m gpu -q -k "not umanoid and not usher and not nt_" > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log
tail -4 $O/tests.log
for rep in 1 2; do
for cfg in "HalfCheetah 65536" "Walker2d 65536" "Hopper 65536"; do
  set -- $cfg
  timeout 600 python bench.py --task $1 --num-envs $2 --no-cpu-baseline 2>>$O/err >> $O/bench.jsonl
done
done
python - <<'PY'
import json
for l in open('gpurun_out/r2y/bench.jsonl'):
    d=json.loads(l); print(d['metric'], '%.4e'%d['value'], 'kernel_ms %.4f'%d['roofline']['kernel_ms'])
PY
