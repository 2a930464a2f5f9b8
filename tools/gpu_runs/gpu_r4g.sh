#!/bin/bash
# round 4, call g: HumanoidStandup's dual matrix through v_mfma_f64_16x16x4_f64 (product build) vs the quad's
# streamed dot products (-DEPA_STANDUP_MFMA=0), same box, interleaved; Humanoid parity tests on the product build first
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_fullsize.py tests/test_gpu_mujoco_golden.py tests/test_gpu_api.py -q -m gpu -x -k "umanoid" 2>&1 | tail -8 > gpurun_out/r4g_tests.log
cat gpurun_out/r4g_tests.log
cp envpool_amd/lib/libenvpool_amd.so /tmp/prod.so
run() {
  timeout 300 python bench.py --task HumanoidStandup --num-envs 65536 --no-cpu-baseline --only-timed --min-time 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', '%.4g' % d['value'], '%.2f ms' % d['roofline']['kernel_ms'])" >> gpurun_out/r4g_mfma_ab.txt
}
for rep in 1 2; do
  cp /tmp/prod.so envpool_amd/lib/libenvpool_amd.so; run mfma
  cp envpool_amd/lib/libenvpool_amd_nomfma.so envpool_amd/lib/libenvpool_amd.so; run nomfma
done
cat gpurun_out/r4g_mfma_ab.txt
