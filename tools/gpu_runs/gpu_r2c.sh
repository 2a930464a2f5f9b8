#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2c
mkdir -p $O
python -m pytest tests -m gpu -x -q -k "ant or Ant or frame_stack or composition or device_path or sharded" > $O/gpu_tests_ant.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests_ant.log
tail -3 $O/gpu_tests_ant.log
python tools/ant_iter_stats.py 32768 > $O/ant_iter_stats_sorted.txt 2>&1; cat $O/ant_iter_stats_sorted.txt
for prec in fp64 fp32; do
  for n in 32768 65536; do
    for srt in 1 0; do
      python bench.py --task Ant --num-envs $n --precision $prec --steps 100 --warmup 20 --no-cpu-baseline --param sort_by_cost=$srt 2>> $O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$prec', $n, 'sort=$srt', '%.3e'%d['value'], 'ms', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],3))" | tee -a $O/bench_ant_sort_ab.txt
    done
  done
done
