#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2m
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x -s > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; grep -E "quad vs|sorted vs|passed|failed|rc=" $O/gpu_tests.log | tail -8
