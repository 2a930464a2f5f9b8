#!/bin/bash
# Round 6, second session: runtime knobs on the cross-stream hand-over latency of the pipelined numpy step
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6t; mkdir -p $O
for rep in 1 2 3; do for knob in none HSA_ENABLE_INTERRUPT=0 HSA_ENABLE_SDMA=0 GPU_MAX_HW_QUEUES=8; do
  if [ $knob = none ]; then pre=""; else pre="env $knob"; fi
  echo "knob=$knob rep=$rep $($pre python tools/numpy_step_ab.py HalfCheetah 65536 32768 6 bind 2>/dev/null | tail -1)"
done; done | tee $O/runtime_knobs_ab.txt
