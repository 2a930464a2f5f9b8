#!/bin/bash
# round 3: flakiness hunt on the final build -- the full GPU suite three times in different orders
# (-p no:randomly is not installed: reverse file order by hand) and every MuJoCo kernel under LDS poisoning
set -u
export TMPDIR=/tmp
O=gpurun_out/r3q
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/run1.log 2>&1; tail -1 $O/run1.log
timeout 900 python -m pytest $(ls tests/test_gpu_*.py tests/test_mjcpu_golden.py tests/test_refbind.py | sort -r) -m gpu -q > $O/run2.log 2>&1; tail -1 $O/run2.log
timeout 900 python -m pytest tests -m gpu -q -x > $O/run3.log 2>&1; tail -1 $O/run3.log
timeout 900 python tools/hum_poison_check.py > $O/poison.log 2>&1; cat $O/poison.log
