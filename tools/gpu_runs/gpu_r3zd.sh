#!/bin/bash
# flakiness check of the Humanoid-family GPU tests on the hybrid-PGS build: five runs of the subset
set -u
export TMPDIR=/tmp
O=gpurun_out/r3zd
mkdir -p $O
for i in 1 2 3 4 5; do
  timeout 600 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_fullsize.py tests/test_gpu_sharded.py -m gpu -q -k "umanoid" 2>&1 | tail -1 | tee -a $O/runs.log
done
