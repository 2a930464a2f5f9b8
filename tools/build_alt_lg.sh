#!/bin/bash
# A/B builds of the lane-group planar TU: envpool_amd/lib/libenvpool_amd_<tag>.so = the product library with
# mujoco_planar_lg.hip compiled with the given flags instead of the Makefile's.  On the GPU box a run
# swaps it in by copying it over libenvpool_amd.so (tools/gpu_runs/*).
#   usage: tools/build_alt_lg.sh <tag> <hipcc flags...>
set -e
TAG=$1; shift
cd "$(dirname "$0")/../envpool_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result "$@" -c mujoco_planar_lg.hip -o build/mujoco_planar_lg_$TAG.o 2>&1 | grep -E "error" -A5 || true
OBJ="build/engine.o build/classic_control.o build/toy_text.o build/mujoco_gym.o build/mujoco_ant.o build/mujoco_pendulum.o build/mujoco_humanoid.o build/mujoco_humanoid_standup.o build/mujoco_humanoid4.o build/mujoco_pusher.o build/atari_post.o build/atari_env.o"  # the product's objects (Makefile: OBJ) minus the ones replaced
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ build/mujoco_planar_lg_$TAG.o -o ../lib/libenvpool_amd_$TAG.so -ldl -lpthread
ls -la ../lib/libenvpool_amd_$TAG.so
