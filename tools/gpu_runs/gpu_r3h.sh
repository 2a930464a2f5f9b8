#!/bin/bash
# round 3: longest-chunk-first dispatch of the lane-group kernel: parity, A/B, profile
set -u
export TMPDIR=/tmp
O=gpurun_out/r3h
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_fullsize.py tests/test_gpu_device_path.py -m gpu -q -s -k "lane_group or spread or teacher_forced_step or walker or device_path or headline or config3" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=|FAILED" $O/tests.log | tail -8
for n in 65536 49152 98304 131072 32768; do for lpt in 1 0; do
  timeout 300 python bench.py --num-envs $n --no-cpu-baseline --min-time 0.5 --param planar_layout=2 --param planar_lpt=$lpt 2>>$O/err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('HalfCheetah n=$n layout=2 lpt=$lpt %.4e ms/step %.4f kernel_ms %.4f'%(d['value'],d['ms_per_step'],d['roofline']['kernel_ms']))" | tee -a $O/sweep.txt
done; done
for n in 8192 16384 32768; do for lpt in 1 0; do
  timeout 300 python bench.py --num-envs $n --no-cpu-baseline --min-time 0.5 --param planar_layout=4 --param planar_lpt=$lpt 2>>$O/err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('HalfCheetah n=$n layout=4 lpt=$lpt %.4e ms/step %.4f kernel_ms %.4f'%(d['value'],d['ms_per_step'],d['roofline']['kernel_ms']))" | tee -a $O/sweep.txt
done; done
for lpt in 1 0; do
  timeout 300 python bench.py --task Walker2d --no-cpu-baseline --min-time 0.5 --param planar_layout=2 --param planar_lpt=$lpt 2>>$O/err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('Walker2d n=65536 layout=2 lpt=$lpt %.4e ms/step %.4f'%(d['value'],d['ms_per_step']))" | tee -a $O/sweep.txt
done
bash tools/profile_bench.sh r3h_lg2_w1_64k --num-envs 65536 --param planar_layout=2 > /dev/null 2>&1; cat gpurun_out/prof_r3h_lg2_w1_64k/summary.md
