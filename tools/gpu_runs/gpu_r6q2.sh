#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r6q; mkdir -p $O
( time timeout 1800 python -m pytest tests/test_gpu_api.py tests/test_gpu_classic_toy.py tests/test_gpu_device_path.py tests/test_gpu_blocking_recv.py tests/test_gpu_step_pipeline.py -m gpu -q -x ) 2>&1 | tail -4
sed -n '/cat > \/tmp\/ab.py/,/^PY$/p' tools/gpu_runs/gpu_r6q.sh | sed '1d;$d' > /tmp/ab.py
python /tmp/ab.py 2>&1 | grep -v amdgpu.ids | grep "rep0\|rep1" | tee $O/small_zero_copy_ab_coherent.txt
