#!/bin/bash
# round 3: lane-group kernel with the table in LDS, persistent waves over a dynamic chunk queue and
# finite termination inside the line search; async batches on several compute streams
set -u
export TMPDIR=/tmp
O=gpurun_out/r3f
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_api.py tests/test_gpu_device_path.py tests/test_gpu_classic_toy.py -m gpu -q -s -k "lane_group or spread or teacher_forced_step or walker or async or device_path" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=|FAILED" $O/tests.log | tail -12
for n in 65536 8192 16384 32768 131072; do for lw in "2 1" "2 2" "4 1" "4 2"; do set -- $lw
  timeout 300 python bench.py --num-envs $n --no-cpu-baseline --min-time 0.5 --param planar_layout=$1 --param planar_waves=$2 2>>$O/err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('HalfCheetah n=$n layout=$1 waves=$2 %.4e ms/step %.4f kernel_ms %.4f'%(d['value'],d['ms_per_step'],d['roofline']['kernel_ms']))" | tee -a $O/sweep.txt
done; done
for lw in "1 2" "2 1" "2 2"; do set -- $lw
  timeout 300 python bench.py --task Walker2d --no-cpu-baseline --min-time 0.5 --param planar_layout=$1 --param planar_waves=$2 2>>$O/err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('Walker2d n=65536 layout=$1 waves=$2 %.4e ms/step %.4f'%(d['value'],d['ms_per_step']))" | tee -a $O/sweep.txt
done
GPU_MAX_HW_QUEUES=8 timeout 600 python tools/bench_async_api.py streams 2>>$O/err | tee $O/async_streams_hwq8.jsonl
timeout 600 python tools/bench_async_api.py streams 2>>$O/err | tee $O/async_streams.jsonl
