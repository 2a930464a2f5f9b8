#!/bin/bash
# round-2 GPU session B: quad-layout Ant kernel -- parity suite, bench sweep, A/B, profile
set -u
export TMPDIR=/tmp
O=gpurun_out/r2b
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
tail -3 $O/gpu_tests.log
for prec in fp64 fp32; do
  for n in 8192 32768 65536 131072; do
    python bench.py --task Ant --num-envs $n --precision $prec --steps 100 --warmup 20 --no-cpu-baseline >> $O/bench_ant.jsonl 2>> $O/bench_ant.err
  done
done
cp envpool_amd/lib/libenvpool_amd.so /tmp/lib_a.so
cp envpool_amd/lib/variant_antw2.so envpool_amd/lib/libenvpool_amd.so
for prec in fp64 fp32; do
  for n in 32768 65536; do
    python bench.py --task Ant --num-envs $n --precision $prec --steps 100 --warmup 20 --no-cpu-baseline >> $O/bench_ant_w2.jsonl 2>> $O/bench_ant.err
  done
done
cp /tmp/lib_a.so envpool_amd/lib/libenvpool_amd.so
python bench.py --steps 100 --warmup 20 --no-cpu-baseline >> $O/bench_cheetah.jsonl 2>> $O/bench_ant.err
bash tools/profile_bench.sh r2b_ant_f64 --task Ant --num-envs 32768 > $O/profile.log 2>&1
cut -c1-400 $O/bench_ant.jsonl | sed 's/"config".*"roofline"/ROOF/' 
cut -c1-300 $O/bench_ant_w2.jsonl | sed 's/"config".*"roofline"/ROOF/'
tail -30 gpurun_out/prof_r2b_ant_f64/summary.md
