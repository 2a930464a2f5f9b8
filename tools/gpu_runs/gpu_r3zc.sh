#!/bin/bash
# Humanoid kernels with the hybrid PGS under LDS poisoning (reads of unwritten LDS show up as NaN / run-to-run
# differences on every run), at the action scale that makes HumanoidStandup take the hybrid form
set -u
export TMPDIR=/tmp
O=gpurun_out/r3zc
mkdir -p $O
bash tools/lds_poison/build.sh > $O/build.log 2>&1
timeout 600 python tools/hum_poison_check.py Humanoid HumanoidStandup > $O/poison.log 2>&1; cat $O/poison.log
