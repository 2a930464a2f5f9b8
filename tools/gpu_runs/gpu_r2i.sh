#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2i
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -5 $O/gpu_tests.log
