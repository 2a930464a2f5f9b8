#!/bin/bash
# Round 5, call s: action upload of the host path in 1 MB slices (DMA of slice i under the host copy of slice i + 1) --
# API / classic / sharded tests, then the numpy-API rates
set -u
export TMPDIR=/tmp
O=gpurun_out/r5s
mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_api.py tests/test_gpu_classic_toy.py tests/test_gpu_device_path.py tests/test_gpu_sharded.py tests/test_gpu_fullsize.py -q ) > $O/gpu_tests.log 2>&1; grep -E "passed|failed|FAILED" $O/gpu_tests.log | tail -5
timeout 600 python tools/bench_numpy_api.py > $O/numpy_api.jsonl 2>>$O/err; cut -c1-200 $O/numpy_api.jsonl
timeout 300 python tools/pcie_probe.py 2>>$O/err | tail -1 > $O/pcie_probe_api_leg.json; cat $O/pcie_probe_api_leg.json
