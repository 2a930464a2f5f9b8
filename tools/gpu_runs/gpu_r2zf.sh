#!/bin/bash
# final spread policy (16..64 envs per wave, only from 16 envs per SIMD up): full GPU suite + a short sweep
set -u
export TMPDIR=/tmp
O=gpurun_out/r2zf
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; grep -E "passed|failed|rc=" $O/gpu_tests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $O/gpu_tests.log 2>&1; tail -1 $O/gpu_tests.log
for n in 8192 16384 32768 65536; do
    timeout 600 python bench.py --num-envs $n --no-cpu-baseline 2>>$O/err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('HalfCheetah n=$n %.4e ms/step %.4f'%(d['value'],d['ms_per_step']))" | tee -a $O/sweep.txt
done
for t in Walker2d Hopper; do
    timeout 600 python bench.py --task $t --num-envs 32768 --no-cpu-baseline 2>>$O/err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t n=32768 %.4e ms/step %.4f'%(d['value'],d['ms_per_step']))" | tee -a $O/sweep.txt
done
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-200 $O/bench_default.json
