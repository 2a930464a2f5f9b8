#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2n
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_api.py -m gpu -x -q -k "humanoid or Humanoid or composition" > $O/gpu_tests_hum.log 2>&1; echo "rc=$?" >> $O/gpu_tests_hum.log; tail -3 $O/gpu_tests_hum.log
bash tools/gpu_runs/gpu_r2l.sh | tail -3
