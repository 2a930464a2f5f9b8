#!/bin/bash
# Round 5, call c: word-at-a-time mt19937 + tiled generator words -- the full GPU suite, the HBM-streaming families'
# rocprofv3 passes again, and the Hopper / Walker2d steady-state traffic
set -u
export TMPDIR=/tmp
O=gpurun_out/r5c
mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q -x ) > $O/gpu_tests.log 2>&1; grep -E "passed|failed|real" $O/gpu_tests.log | tail -4
bash tools/profile_families.sh r5c_families --families CartPole,MountainCar,Pendulum,Acrobot,FrozenLake,NChain,Blackjack,Catch,Taxi > $O/families.log 2>&1
tail -24 $O/families.log
bash tools/profile_bench.sh r5c_hopper_lg1 --task Hopper > $O/hopper.log 2>&1; grep -A3 "timed window" $O/hopper.log | head -4; grep -E "FETCH_SIZE|WRITE_SIZE" $O/hopper.log
bash tools/profile_bench.sh r5c_walker_lg2 --task Walker2d > $O/walker.log 2>&1; grep -A3 "timed window" $O/walker.log | head -4; grep -E "FETCH_SIZE|WRITE_SIZE" $O/walker.log
