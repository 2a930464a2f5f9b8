#!/bin/bash
# Ant profiles after the DPP change (kernel trace + PMC passes), for profiles/pmc.json
set -u
export TMPDIR=/tmp
O=gpurun_out/r2v
mkdir -p $O
bash tools/profile_bench.sh r2v_ant32k_f64 --task Ant --num-envs 32768 > $O/p2.log 2>&1
bash tools/profile_bench.sh r2v_ant64k_f64 --task Ant --num-envs 65536 > $O/p3.log 2>&1
bash tools/profile_bench.sh r2v_ant64k_f32 --task Ant --num-envs 65536 --precision fp32 > $O/p4.log 2>&1
grep -E "FETCH_SIZE|WRITE_SIZE|AntStepKernel.*\| [0-9]+ \|" gpurun_out/prof_r2v_*/summary.md
