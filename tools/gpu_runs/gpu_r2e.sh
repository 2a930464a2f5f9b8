#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2e
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_atari_env.py tests/test_gpu_atari_post.py -x -q > $O/atari_tests.log 2>&1; echo "rc=$?" >> $O/atari_tests.log
tail -25 $O/atari_tests.log
