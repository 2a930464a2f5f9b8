#!/bin/bash
# Diagnostic build: envpool_amd/lib/libenvpool_amd_anttimers.so = the product library with the Ant TU's stage
# timers compiled in (-DEPA_ANT_TIMERS: mj_ant4.hip.h EPA_ANT_TICK, mujoco_ant.hip epa_debug_ant_timers).
# tools/ant_stage_timers.py loads it through ENVPOOL_AMD_LIB.
set -e
cd "$(dirname "$0")/../envpool_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -mllvm -disable-machine-licm \
  -mllvm -amdgpu-spill-sgpr-to-vgpr=false -fno-slp-vectorize -DEPA_ANT_TIMERS -c mujoco_ant.hip -o build/mujoco_ant_timers.o 2>&1 | grep -E "error" -A5 || true
OBJ="build/engine.o build/classic_control.o build/toy_text.o build/mujoco_gym.o build/mujoco_planar_lg.o build/mujoco_pendulum.o build/mujoco_humanoid.o build/mujoco_humanoid_standup.o build/mujoco_humanoid4.o build/mujoco_pusher.o build/atari_post.o build/atari_env.o"  # the product's objects (Makefile: OBJ) minus the ones replaced
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ build/mujoco_ant_timers.o -o ../lib/libenvpool_amd_anttimers.so -ldl -lpthread
ls -la ../lib/libenvpool_amd_anttimers.so
