#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2d
mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-1800 $O/bench_default.json
bash tools/profile_bench.sh r2d_ant32k_f64 --task Ant --num-envs 32768 > $O/p1.log 2>&1
bash tools/profile_bench.sh r2d_ant64k_f64 --task Ant --num-envs 65536 > $O/p2.log 2>&1
bash tools/profile_bench.sh r2d_cheetah_f64 > $O/p3.log 2>&1
bash tools/profile_bench.sh r2d_ant64k_f32 --task Ant --num-envs 65536 --precision fp32 > $O/p4.log 2>&1
head -8 gpurun_out/prof_r2d_*/summary.md
