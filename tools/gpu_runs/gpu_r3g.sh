#!/bin/bash
# round 3: batch-independence fix of the lane-group solver; stream-affine async batches
set -u
export TMPDIR=/tmp
O=gpurun_out/r3g
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_api.py tests/test_gpu_device_path.py tests/test_gpu_classic_toy.py tests/test_gpu_fullsize.py -m gpu -q -s -k "lane_group or spread or teacher_forced_step or walker or async or device_path or headline or config3" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=|FAILED" $O/tests.log | tail -12
timeout 600 python tools/bench_async_api.py streams 2>>$O/err | tee $O/async_streams.jsonl
GPU_MAX_HW_QUEUES=8 timeout 600 python tools/bench_async_api.py streams 2>>$O/err | tee $O/async_streams_hwq8.jsonl
