#!/bin/bash
# A/B builds of the Humanoid quad TU: envpool_amd/lib/libenvpool_amd_<tag>.so = the product library with
# mujoco_humanoid4.hip compiled with extra flags (e.g. -DEPA_STANDUP_REGROWS=20).  On the GPU box a run
# swaps it in by copying it over libenvpool_amd.so (tools/gpu_runs/*).
#   usage: tools/build_alt_hum4.sh <tag> <extra hipcc flags...>
set -e
TAG=$1; shift
cd "$(dirname "$0")/../envpool_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -mllvm -disable-machine-licm \
  ${DPPFLAG-} "$@" -c mujoco_humanoid4.hip -o build/alt_humanoid4_$TAG.o 2>&1 | grep -E "error" -A5 || true
OBJ=$(ls build/*.o | grep -v "_trace.o" | grep -v "alt_" | grep -v "mujoco_planar_lg_" | grep -v "build/mujoco_humanoid4.o" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ build/alt_humanoid4_$TAG.o -o ../lib/libenvpool_amd_$TAG.so -ldl -lpthread
ls -la ../lib/libenvpool_amd_$TAG.so
