#!/bin/bash
# kernel durations (rocprofv3 kernel trace) of the classic step kernels at num_envs = 65536, classic_early = 0 / 1
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r6o; mkdir -p $O
for rep in 1 2; do for e in 0 1; do
  cd /tmp; rm -rf /tmp/ce$e
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ce$e -o t -- python $R/tools/bench_families.py --families CartPole,Pendulum,MountainCar,MountainCarContinuous,Acrobot --no-atari --big 0 --warmup 700 --steps 400 --param classic_early=$e > /dev/null 2>&1
  cd $R
  python - $e $rep <<'PY'
import csv, glob, sys
e, rep = sys.argv[1], sys.argv[2]
kinds = {0: "CartPole", 1: "Pendulum", 2: "MountainCar", 3: "MountainCarContinuous", 4: "Acrobot"}
for f in glob.glob(f"/tmp/ce{e}/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ClassicStepKernel" in r["Name"]:
            print(f"classic_early={e} rep{rep} {r['Name'][:80]:80s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:7.2f} us  min {float(r['MinNs'])/1e3:6.2f}")
PY
done; done | tee $O/classic_early_kernel_trace.txt
