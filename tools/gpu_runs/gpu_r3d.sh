#!/bin/bash
# round 3: first run of the lane-group planar kernel (mujoco_planar_lg.hip): parity of every variant,
# the Humanoid LDS fix under poisoning, then bench A/B over layout x register budget x batch size
set -u
export TMPDIR=/tmp
O=gpurun_out/r3d
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mujoco.py -m gpu -q -s -k "lane_group or spread or teacher_forced_step or walker" > $O/tests_lg.log 2>&1; echo "rc=$?" >> $O/tests_lg.log; grep -E "passed|failed|rc=|layout" $O/tests_lg.log | tail -30
timeout 600 python tools/hum_poison_check.py Humanoid HumanoidStandup HalfCheetah Walker2d > $O/poison.log 2>&1; cat $O/poison.log
for n in 65536 8192 16384 32768 131072; do for lw in "1 2" "2 2" "2 1" "4 2" "4 1"; do set -- $lw
  timeout 300 python bench.py --num-envs $n --no-cpu-baseline --min-time 0.5 --param planar_layout=$1 --param planar_waves=$2 2>>$O/err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('HalfCheetah n=$n layout=$1 waves=$2 %.4e ms/step %.4f kernel_ms %.4f'%(d['value'],d['ms_per_step'],d['roofline']['kernel_ms']))" | tee -a $O/sweep.txt
done; done
for lw in "1 2" "2 2" "2 1" "4 2"; do set -- $lw
  timeout 300 python bench.py --task Walker2d --no-cpu-baseline --min-time 0.5 --param planar_layout=$1 --param planar_waves=$2 2>>$O/err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('Walker2d n=65536 layout=$1 waves=$2 %.4e ms/step %.4f'%(d['value'],d['ms_per_step']))" | tee -a $O/sweep.txt
done
