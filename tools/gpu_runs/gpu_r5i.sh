#!/bin/bash
# Round 5, call i: reset draws as one burst of generator words (Mt19937::NextWords) -- classic-control parity tests,
# then the steady-state A/B against the build without it (libenvpool_amd_noburst.so = the parent commit's sources)
set -u
export TMPDIR=/tmp
O=gpurun_out/r5i
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_classic_toy.py tests/test_gpu_fullsize.py tests/test_gpu_api.py -q -x ) > $O/gpu_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tests.log | tail -3
FAM="--families CartPole,Pendulum,MountainCar,Acrobot --no-atari --warmup 700 --steps 200 --big 4194304"
cp envpool_amd/lib/libenvpool_amd.so /tmp/new.so
echo "== burst draws (this commit)" > $O/burst_ab.txt
python tools/bench_families.py $FAM 2>>$O/err | grep "^|" >> $O/burst_ab.txt
cp envpool_amd/lib/libenvpool_amd_noburst.so envpool_amd/lib/libenvpool_amd.so
echo "== one word per Next() (parent commit)" >> $O/burst_ab.txt
python tools/bench_families.py $FAM 2>>$O/err | grep "^|" >> $O/burst_ab.txt
cp /tmp/new.so envpool_amd/lib/libenvpool_amd.so
cat $O/burst_ab.txt
tail -3 $O/err
