#!/bin/bash
# planar kernel: small batches spread over all SIMDs (16..64 envs per wave) -- parity + num_envs sweep, A/B
set -u
export TMPDIR=/tmp
O=gpurun_out/r2ze
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_fullsize.py tests/test_gpu_api.py tests/test_gpu_device_path.py tests/test_gpu_sharded.py -m gpu -q -k "not umanoid" > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log
tail -3 $O/tests.log
for n in 1024 4096 8192 16384 24576 32768 49152 65536; do
  for sp in 1 0; do
    timeout 600 python bench.py --num-envs $n --no-cpu-baseline --param planar_spread=$sp 2>>$O/err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('HalfCheetah n=$n spread=$sp %.4e ms/step %.4f'%(d['value'],d['ms_per_step']))" | tee -a $O/sweep.txt
  done
done
for t in Walker2d Hopper; do for n in 8192 32768; do for sp in 1 0; do
    timeout 600 python bench.py --task $t --num-envs $n --no-cpu-baseline --param planar_spread=$sp 2>>$O/err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t n=$n spread=$sp %.4e ms/step %.4f'%(d['value'],d['ms_per_step']))" | tee -a $O/sweep.txt
done; done; done
for sp in 1 0; do
timeout 300 python bench.py --num-envs 8192 --precision fp32 --no-cpu-baseline --param planar_spread=$sp 2>>$O/err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('HalfCheetah fp32 n=8192 spread=$sp %.4e ms/step %.4f'%(d['value'],d['ms_per_step']))" | tee -a $O/sweep.txt
done
