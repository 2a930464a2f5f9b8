#!/bin/bash
# round 3: the Humanoid run-to-run determinism test failed once (NaN in one of two identical pools):
# repeat it, and run every MuJoCo kernel with the CUs' LDS poisoned before each step
set -u
export TMPDIR=/tmp
O=gpurun_out/r3c
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "classic_config2" > $O/fullsize.log 2>&1; tail -3 $O/fullsize.log
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_gpu_api.py -m gpu -q -k "determinism" 2>&1 | tail -1; done | tee $O/determinism_repeat.log
timeout 900 python tools/hum_poison_check.py > $O/poison.log 2>&1; cat $O/poison.log
