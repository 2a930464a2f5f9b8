#!/bin/bash
# Round 6, second session: does the numpy step depend on the socket the calling thread runs on?
export TMPDIR=/tmp
O=gpurun_out/r6x; mkdir -p $O
{
lscpu | grep -E "NUMA|Socket|Model name|^CPU\(s\)"
for d in /sys/class/drm/card*/device; do echo "$d numa_node=$(cat $d/numa_node 2>/dev/null) vendor=$(cat $d/vendor 2>/dev/null)"; done
which numactl
for rep in 1 2 3; do
  for cpus in 0-63 64-127 none; do
    if [ $cpus = none ]; then pre=""; else pre="taskset -c $cpus"; fi
    echo "cpus=$cpus rep=$rep $($pre python tools/numpy_step_ab.py HalfCheetah 65536 32768 6 2>/dev/null | tail -1)"
  done
done
} 2>&1 | tee $O/numa_probe.txt
