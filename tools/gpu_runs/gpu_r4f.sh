#!/bin/bash
# round 4, call f: A/B of the software-pipelined slot loops of the lane-group kernel (prefetch = product build,
# noprefetch = -DEPA_LG_NO_PREFETCH), same box, alternating; then the parity tests of the planar families
mkdir -p gpurun_out
cp envpool_amd/lib/libenvpool_amd.so /tmp/prod.so
run() {  # tag task n
  timeout 300 python bench.py --task $2 --num-envs $3 --no-cpu-baseline --min-time 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', '$2', $3, '%.4g' % d['value'], '%.1f us' % (d['roofline']['kernel_ms']*1e3), 'async %.4g' % d['async_mode']['value'])" >> gpurun_out/r4f_prefetch_ab.txt
}
for rep in 1 2; do
  cp /tmp/prod.so envpool_amd/lib/libenvpool_amd.so
  run prefetch HalfCheetah 65536; run prefetch HalfCheetah 8192; run prefetch Walker2d 65536; run prefetch Hopper 65536
  cp envpool_amd/lib/libenvpool_amd_noprefetch.so envpool_amd/lib/libenvpool_amd.so
  run noprefetch HalfCheetah 65536; run noprefetch HalfCheetah 8192; run noprefetch Walker2d 65536; run noprefetch Hopper 65536
done
cp /tmp/prod.so envpool_amd/lib/libenvpool_amd.so
timeout 1200 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_fullsize.py tests/test_gpu_mujoco_golden.py -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r4f_tests.log
cat gpurun_out/r4f_prefetch_ab.txt gpurun_out/r4f_tests.log
