#!/bin/bash
# Round 5, call e: (1) steady-state A/B of the lazily regenerated mt19937 against the block-wise one of rounds 1-4
# (library built from the parent commit), 700 warm-up steps so that every env is past its first 624 words;
# (2) the chunk schedule + the cost of the reset branch for Hopper / Walker2d / HalfCheetah
set -u
export TMPDIR=/tmp
O=gpurun_out/r5e
mkdir -p $O
FAM="--families CartPole,Acrobot,FrozenLake,NChain,CliffWalking,Blackjack,Taxi,InvertedPendulum,InvertedDoublePendulum --no-atari --warmup 700 --steps 200 --big 4194304"
cp envpool_amd/lib/libenvpool_amd.so /tmp/new.so
echo "== lazy regeneration (this commit)" > $O/mt_ab.txt
python tools/bench_families.py $FAM 2>>$O/err | grep "^|" >> $O/mt_ab.txt
cp envpool_amd/lib/libenvpool_amd_r4mt.so envpool_amd/lib/libenvpool_amd.so
echo "== block-wise twist (parent commit)" >> $O/mt_ab.txt
python tools/bench_families.py $FAM 2>>$O/err | grep "^|" >> $O/mt_ab.txt
cat $O/mt_ab.txt
cp envpool_amd/lib/libenvpool_amd_sched.so envpool_amd/lib/libenvpool_amd.so
for cfg in "Hopper 65536" "Walker2d 65536" "HalfCheetah 65536"; do
  set -- $cfg
  timeout 300 python tools/lg_sched_trace.py $1 $2 20 >> $O/lg_sched_trace.txt 2>> $O/err
done
grep -E "N=|span|busy|  mean|reset_branch|chunks_with" $O/lg_sched_trace.txt
cp /tmp/new.so envpool_amd/lib/libenvpool_amd.so
tail -3 $O/err
