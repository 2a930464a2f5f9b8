#!/bin/bash
# default bench command under rocprofv3 (kernel trace + PMC passes) on the final build
set -u
export TMPDIR=/tmp
O=gpurun_out/r2za
mkdir -p $O
bash tools/profile_bench.sh r2z_cheetah_f64 > $O/p.log 2>&1
sed -n 1,16p gpurun_out/prof_r2z_cheetah_f64/summary.md
