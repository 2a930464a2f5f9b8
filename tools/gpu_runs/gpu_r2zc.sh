#!/bin/bash
# Humanoid / HumanoidStandup under rocprofv3 with the action-space draw (the bench's default command)
set -u
export TMPDIR=/tmp
O=gpurun_out/r2zc
mkdir -p $O
bash tools/profile_bench.sh r2zc_humanoid4 --task Humanoid --num-envs 65536 > $O/p1.log 2>&1
bash tools/profile_bench.sh r2zc_standup4 --task HumanoidStandup --num-envs 65536 > $O/p2.log 2>&1
grep -E "FETCH_SIZE|WRITE_SIZE|timed window|launches [0-9]|Humanoid4StepKernel<double>.*\| [0-9]+ \|" gpurun_out/prof_r2zc_*/summary.md
