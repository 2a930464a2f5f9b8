#!/bin/bash
# round 3, first call: re-probe for mujoco / opencv, full GPU suite with the new config-sized parity
# tests, baseline bench lines (default, N=8192) before the planar kernel is re-laid out
set -u
export TMPDIR=/tmp
O=gpurun_out/r3a
mkdir -p $O
bash tools/probe_refs.sh > $O/probe_gpu_box.log 2>&1; grep -E "FOUND|reachable" $O/probe_gpu_box.log | head
timeout 1500 python -m pytest tests -m gpu -q -s --durations=15 > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; grep -E "passed|failed|rc=|Error|error" $O/gpu_tests.log | tail -8
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json
timeout 300 python bench.py --num-envs 8192 --no-cpu-baseline > $O/bench_8192.json 2>> $O/err; cut -c1-300 $O/bench_8192.json
