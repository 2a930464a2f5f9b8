#!/bin/bash
# kernel durations (rocprofv3 kernel trace) of the toy_text / chain step kernels at num_envs = 65536
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r6o; mkdir -p $O
cd /tmp; rm -rf /tmp/tt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tt -o t -- python $R/tools/bench_families.py --families ${FAMS:-Catch,FrozenLake,Taxi,NChain,CliffWalking,Blackjack,InvertedPendulum,InvertedDoublePendulum,Reacher,Swimmer} --no-atari --big ${BIG:-0} --warmup 700 --steps 400 > /dev/null 2>&1
cd $R
python - <<'PY' | tee $O/${OUTNAME:-toy_kernel_trace.txt}
import csv, glob
for f in glob.glob("/tmp/tt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "StepKernel" in r["Name"]:
            print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:7.2f} us  min {float(r['MinNs'])/1e3:6.2f}")
PY
