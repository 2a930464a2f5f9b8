#!/bin/bash
# Interleaved A/B of alternative builds of the library on one box: tools/lib_ab.sh "<lib tags>" "<task:num_envs ...>" [reps]
# (tag "product" = libenvpool_amd.so, else libenvpool_amd_<tag>.so); one line per run -> stdout
LIBS=$1; CFGS=$2; REPS=${3:-2}
for rep in $(seq $REPS); do for cfg in $CFGS; do for tag in $LIBS; do
  task=${cfg%%:*}; n=${cfg##*:}
  lib=$PWD/envpool_amd/lib/libenvpool_amd$([ $tag = product ] || echo _$tag).so
  ENVPOOL_AMD_LIB=$lib python bench.py --task $task --num-envs $n --only-timed --no-cpu-baseline --min-time 2 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', '$task', $n, 'rep$rep', '%.4e' % d['value'], '%.4f ms' % d['ms_per_step'])"
done; done; done
