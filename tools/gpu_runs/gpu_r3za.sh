#!/bin/bash
# Async-mode pools after the layout rule change (lanes per env from 4 x batch_size; no partly filled waves):
# the async / device-path tests, the streams table of tools/bench_async_api.py, and the default bench line.
set -u
export TMPDIR=/tmp
O=gpurun_out/r3za
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_device_path.py tests/test_gpu_mujoco.py -m gpu -q > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; grep -E "passed|failed|rc=|FAILED" $O/gpu_tests.log | tail -6
timeout 600 python tools/bench_async_api.py streams > $O/async_streams.jsonl 2>$O/err; cat $O/async_streams.jsonl | cut -c1-400
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2>>$O/err; python -c "
import json; d=json.load(open('$O/bench_default.json')); print('%.3e'%d['value'], d['roofline']['kernel_ms'], d.get('async_mode'))"
