#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2g
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "ant or Ant or frame_stack or composition" > $O/gpu_tests_ant.log 2>&1; echo "rc=$?" >> $O/gpu_tests_ant.log; tail -2 $O/gpu_tests_ant.log
for prec in fp64 fp32; do for n in 32768 65536; do
  python bench.py --task Ant --num-envs $n --precision $prec --steps 100 --warmup 20 --no-cpu-baseline 2>>$O/err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$prec', $n, '%.3e'%d['value'], 'kernel_ms', round(d['roofline']['kernel_ms'],3))" | tee -a $O/bench_ant.txt
done; done
