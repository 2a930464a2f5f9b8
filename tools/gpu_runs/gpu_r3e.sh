#!/bin/bash
# round 3: PMC profile of the lane-group kernel (2 lanes per env) at one wave per SIMD (N=32768, register
# budget 512) and at the headline size with two waves per SIMD (N=65536, budget 256)
set -u
export TMPDIR=/tmp
bash tools/profile_bench.sh r3e_lg2_w1_32k --num-envs 32768 --param planar_layout=2 --param planar_waves=1 > /dev/null 2>&1
bash tools/profile_bench.sh r3e_lg2_w2_64k --num-envs 65536 --param planar_layout=2 --param planar_waves=2 > /dev/null 2>&1
bash tools/profile_bench.sh r3e_lg4_w1_16k --num-envs 16384 --param planar_layout=4 --param planar_waves=1 > /dev/null 2>&1
for t in r3e_lg2_w1_32k r3e_lg2_w2_64k r3e_lg4_w1_16k; do cat gpurun_out/prof_$t/summary.md; done
