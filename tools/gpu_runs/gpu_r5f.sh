#!/bin/bash
# Round 5, call f: generator-word layout per family (mt_tile 1 vs 16, lazy regeneration), the block-wise twist over a
# window that holds whole twist periods, and the split chunk = mj_steps + the rest for Hopper / Walker2d / HalfCheetah
set -u
export TMPDIR=/tmp
O=gpurun_out/r5f
mkdir -p $O
cp envpool_amd/lib/libenvpool_amd.so /tmp/new.so
FAM="--families CartPole,Acrobot,FrozenLake,Taxi,Blackjack,InvertedPendulum --no-atari --warmup 700 --steps 200 --num-envs 65536 --big 4194304"
for tile in 1 16; do
  echo "== lazy regeneration, mt_tile=$tile" >> $O/mt_tile_ab.txt
  python tools/bench_families.py $FAM --param mt_tile=$tile 2>>$O/err | grep "^|" >> $O/mt_tile_ab.txt
done
cp envpool_amd/lib/libenvpool_amd_r4mt.so envpool_amd/lib/libenvpool_amd.so
echo "== block-wise twist (parent commit), 1248 timed steps = whole twist periods of the lockstep families" >> $O/mt_tile_ab.txt
python tools/bench_families.py --families NChain,CliffWalking,FrozenLake,Taxi --no-atari --warmup 100 --steps 1248 --num-envs 65536 --big 4194304 2>>$O/err | grep "^|" >> $O/mt_tile_ab.txt
cp /tmp/new.so envpool_amd/lib/libenvpool_amd.so
echo "== lazy regeneration (defaults), 1248 timed steps" >> $O/mt_tile_ab.txt
python tools/bench_families.py --families NChain,CliffWalking,FrozenLake,Taxi --no-atari --warmup 100 --steps 1248 --num-envs 65536 --big 4194304 2>>$O/err | grep "^|" >> $O/mt_tile_ab.txt
cat $O/mt_tile_ab.txt
cp envpool_amd/lib/libenvpool_amd_sched.so envpool_amd/lib/libenvpool_amd.so
for cfg in "Hopper 65536" "Walker2d 65536" "HalfCheetah 65536"; do
  set -- $cfg
  timeout 300 python tools/lg_sched_trace.py $1 $2 20 >> $O/lg_sched_trace.txt 2>> $O/err
done
grep -E "N=|span|busy|  mean|reset_branch|chunks_with|mj_steps" $O/lg_sched_trace.txt
cp /tmp/new.so envpool_amd/lib/libenvpool_amd.so
tail -3 $O/err
