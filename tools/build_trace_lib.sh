#!/bin/bash
# Diagnostic build: envpool_amd/lib/libenvpool_amd_trace.so = the product library with the
# per-wave trace code of the Ant / planar kernels compiled in (-DEPA_WAVE_TRACE).  Use it by
# copying it over libenvpool_amd.so on the GPU box (tools/ant_trace_stats.py,
# tools/planar_trace_stats.py, tools/ant_iter_stats.py read what it records).
set -e
cd "$(dirname "$0")/../envpool_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -mllvm -disable-machine-licm -mllvm -amdgpu-spill-sgpr-to-vgpr=false -DEPA_WAVE_TRACE"
/opt/rocm/bin/hipcc $F -c mujoco_gym.hip -o build/mujoco_gym_trace.o &
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -c mujoco_ant.hip -o build/mujoco_ant_trace.o &
# "hum_debug" (stage switches / solver statistics of the Humanoid quad kernel) is accepted by this build only
G="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -mllvm -disable-machine-licm -DEPA_HUM_DEBUG"
/opt/rocm/bin/hipcc $G -c mujoco_humanoid.hip -o build/mujoco_humanoid_trace.o &
/opt/rocm/bin/hipcc $G -DEPA_HUM_STANDUP_TU -c mujoco_humanoid.hip -o build/mujoco_humanoid_standup_trace.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/engine.o build/classic_control.o build/toy_text.o \
  build/mujoco_gym_trace.o build/mujoco_ant_trace.o build/mujoco_pendulum.o build/mujoco_humanoid_trace.o \
  build/mujoco_humanoid_standup_trace.o build/mujoco_humanoid4.o build/mujoco_pusher.o build/atari_post.o build/atari_env.o -o ../lib/libenvpool_amd_trace.so
