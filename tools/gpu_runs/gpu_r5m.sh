#!/bin/bash
# Round 5, call m: the same soak on the exact line search (-DEPA_LG_LS_MAX=24: every trip searches exactly, as rounds 1-4)
set -u
export TMPDIR=/tmp
O=gpurun_out/r5m
mkdir -p $O
cp envpool_amd/lib/libenvpool_amd.so /tmp/base.so
cp envpool_amd/lib/libenvpool_amd_ls24.so envpool_amd/lib/libenvpool_amd.so
timeout 900 python tools/lg_iter_soak.py 400 > $O/iter_soak_exact.txt 2>>$O/err; cut -c1-600 $O/iter_soak_exact.txt
cp /tmp/base.so envpool_amd/lib/libenvpool_amd.so
