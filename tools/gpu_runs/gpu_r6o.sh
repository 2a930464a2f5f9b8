#!/bin/bash
# Round 6, second session: classic_control step kernel with every input read in front of the reset branch (classic_early)
export TMPDIR=/tmp
O=gpurun_out/r6o; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_classic_toy.py tests/test_gpu_api.py tests/test_gpu_fullsize.py -m gpu -q -x -k "classic or Classic or CartPole or cartpole or determinism or autoreset or config2" ) 2>&1 | tail -3
for rep in 1 2; do for e in 0 1; do
  echo "== classic_early=$e rep$rep"
  python tools/bench_families.py --families CartPole,Pendulum,MountainCar,MountainCarContinuous,Acrobot --no-atari --warmup 700 --param classic_early=$e 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-22s N=%-8d kernel %.2f us  step (step_device window) %.2f us' % (d['family'], d['num_envs'], d['kernel_us'], d['step_us_step_device']))"
done; done | tee $O/classic_early_ab.txt
