#!/bin/bash
# Round 6, second session: the unit queue on the planar lane-group kernel (planar_sub): identity test, then A/B
set -u
export TMPDIR=/tmp
O=gpurun_out/r6u
mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_mujoco.py -m gpu -q -x -k "unit_queue or lane_group_batch_independent" ) > $O/unit_tests.log 2>&1
echo "rc=$?" >> $O/unit_tests.log; grep -E "passed|failed|rc=|FAILED|Error|assert" $O/unit_tests.log | tail -12
B() { timeout 120 python bench.py --only-timed --no-cpu-baseline --min-time 2 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4e' % d['value'], '%.4f ms' % d['ms_per_step'])"; }
for rep in 1 2; do
for cfg in "HalfCheetah 65536" "HalfCheetah 131072" "HalfCheetah 32768" "Walker2d 65536" "Hopper 131072" "Hopper 65536"; do
  set -- $cfg
  for s in 0 1 2 3; do
    echo "$1 $2 sub=$s rep$rep $(B --task $1 --num-envs $2 --param planar_sub=$s)"
  done
done; done | tee $O/planar_sub_ab.txt
