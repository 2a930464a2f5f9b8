#!/bin/bash
# TEST FIXTURE: envpool_amd/lib/libenvpool_amd_licm.so = the product library with the MuJoCo translation units compiled
# WITHOUT the Makefile's MJFLAGS (machine LICM on, SGPR spills into VGPR lanes) -- the build the Makefile warns about.
# tests/test_gpu_selftest.py loads it through ENVPOOL_AMD_LIB and expects the load-time self-test to refuse it (or, if
# this compiler happens to get it right, to pass it AND agree with the product library).
set -e
cd "$(dirname "$0")/../envpool_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result"
for tu in mujoco_gym mujoco_planar_lg mujoco_pusher; do
  /opt/rocm/bin/hipcc $F -c $tu.hip -o build/${tu}_licm.o 2>&1 | grep -E "error" -A5 || true &
done
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -c mujoco_ant.hip -o build/mujoco_ant_licm.o 2>&1 | grep -E "error" -A5 || true &
wait
OBJ="build/engine.o build/classic_control.o build/toy_text.o build/mujoco_pendulum.o build/mujoco_humanoid.o build/mujoco_humanoid_standup.o build/mujoco_humanoid4.o build/atari_post.o build/atari_env.o"  # the product's objects (Makefile: OBJ) minus the ones replaced
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ build/mujoco_gym_licm.o build/mujoco_planar_lg_licm.o build/mujoco_pusher_licm.o build/mujoco_ant_licm.o -o ../lib/libenvpool_amd_licm.so -ldl -lpthread
ls -la ../lib/libenvpool_amd_licm.so
