#!/bin/bash
# Round 6, second session: the chunk-start latency changes of the planar lane-group kernel (TouchChunk): tests, then A/B
# against the build of HEAD's source (libenvpool_amd_base.so)
set -u
export TMPDIR=/tmp
O=gpurun_out/r6w
mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_fullsize.py tests/test_gpu_selftest.py tests/test_gpu_mujoco_golden.py -m gpu -q -x ) > $O/tests.log 2>&1
echo "rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=|FAILED|Error|assert" $O/tests.log | tail -12
bash tools/lib_ab.sh "base product" "HalfCheetah:65536 HalfCheetah:131072 HalfCheetah:32768 HalfCheetah:8192 Walker2d:65536 Hopper:65536" 2 | tee $O/touch_ab.txt
