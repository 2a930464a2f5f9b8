#!/bin/bash
# Round-2 final measurement pass: default bench line, bench lines of the other MuJoCo kernels,
# rocprofv3 kernel trace + PMC passes (tools/profile_bench.sh) and the per-family table.
set -u
export TMPDIR=/tmp
O=gpurun_out/r2p
mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-600 $O/bench_default.json
for cfg in "Ant 32768" "Ant 65536" "Humanoid 65536" "HumanoidStandup 65536" "Pusher 65536" "Walker2d 65536" "Hopper 65536"; do
  set -- $cfg
  timeout 600 python bench.py --task $1 --num-envs $2 --no-cpu-baseline 2>>$O/err >> $O/bench.jsonl
done
timeout 300 python bench.py --task Ant --num-envs 65536 --precision fp32 --no-cpu-baseline 2>>$O/err >> $O/bench.jsonl
timeout 300 python bench.py --precision fp32 --no-cpu-baseline 2>>$O/err >> $O/bench.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r2p/bench.jsonl'):
    d=json.loads(l); print(d['metric'], d['dtype'], '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'])
PY
bash tools/profile_bench.sh r2p_cheetah_f64 > $O/p1.log 2>&1
bash tools/profile_bench.sh r2p_ant32k_f64 --task Ant --num-envs 32768 > $O/p2.log 2>&1
bash tools/profile_bench.sh r2p_ant64k_f64 --task Ant --num-envs 65536 > $O/p3.log 2>&1
bash tools/profile_bench.sh r2p_ant64k_f32 --task Ant --num-envs 65536 --precision fp32 > $O/p4.log 2>&1
bash tools/profile_bench.sh r2p_humanoid4 --task Humanoid --num-envs 65536 > $O/p5.log 2>&1
bash tools/profile_bench.sh r2p_standup4 --task HumanoidStandup --num-envs 65536 > $O/p6.log 2>&1
bash tools/profile_bench.sh r2p_pusher --task Pusher --num-envs 65536 > $O/p7.log 2>&1
head -6 gpurun_out/prof_r2p_*/summary.md | grep -v "^$" | head -60
