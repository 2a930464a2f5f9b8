#!/bin/bash
# Humanoid quad kernel, final build: kernel trace + PMC passes for profiles/pmc.json
set -u
export TMPDIR=/tmp
O=gpurun_out/r2x
mkdir -p $O
bash tools/profile_bench.sh r2x_humanoid4 --task Humanoid --num-envs 65536 > $O/p.log 2>&1
grep -E "FETCH_SIZE|WRITE_SIZE|SQ_WAIT_ANY|SQ_WAVE_CYCLES|Humanoid4StepKernel<double>.*\| [0-9]+ \|" gpurun_out/prof_r2x_*/summary.md
