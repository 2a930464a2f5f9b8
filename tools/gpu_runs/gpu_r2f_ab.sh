#!/bin/bash
# A/B the variant libraries: HalfCheetah default + Walker2d; correctness guard = planar parity tests
cp envpool_amd/lib/libenvpool_amd.so /tmp/lib_base.so
mkdir -p gpurun_out/r2f
for v in base noslp maxilp nopostsched o2 licm unroll0 memclause nomachinesink; do
  if [ $v = base ]; then cp /tmp/lib_base.so envpool_amd/lib/libenvpool_amd.so; else cp envpool_amd/lib/var_$v.so envpool_amd/lib/libenvpool_amd.so; fi
  c=$(python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4e kernel_ms %.4f'%(d['value'], d['roofline']['kernel_ms']))")
  w=$(python bench.py --task Walker2d --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4e'%d['value'])")
  t=$(timeout 300 python -m pytest tests/test_gpu_mujoco.py -x -q -k "teacher_forced_step or determin" 2>&1 | tail -1)
  echo "$v cheetah $c walker $w tests: $t" | tee -a gpurun_out/r2f/planar_flag_ab.txt
done
cp /tmp/lib_base.so envpool_amd/lib/libenvpool_amd.so
