#!/bin/bash
# round 4, call e: the Hopper on the lane-group kernel (group of one lane): parity tests, then A/B of the
# two Hopper kernels through bench.py (no CPU baseline), N = 65536 / 8192
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mujoco.py -q -m gpu -x -k "hopper or Hopper" 2>&1 | tail -12 > gpurun_out/r4e_hopper_tests.log
timeout 600 python -m pytest tests/test_gpu_mujoco_golden.py tests/test_gpu_api.py -q -m gpu -x 2>&1 | tail -5 >> gpurun_out/r4e_hopper_tests.log
for L in 0 1; do
  for N in 65536 8192; do
    timeout 300 python bench.py --task Hopper --num-envs $N --no-cpu-baseline --min-time 2 --param planar_layout=$L >> gpurun_out/r4e_hopper_bench.jsonl 2>> gpurun_out/r4e_hopper_bench.err
  done
done
cat gpurun_out/r4e_hopper_tests.log
python - <<'PY'
import json
for l in open("gpurun_out/r4e_hopper_bench.jsonl"):
    d = json.loads(l)
    print(d["config"]["params"], d["config"]["num_envs_per_gpu"], f'{d["value"]:.4g} env-steps/s', f'{d["roofline"]["kernel_ms"]*1e3:.1f} us',
          "async", f'{d["async_mode"]["value"]:.4g}')
PY
