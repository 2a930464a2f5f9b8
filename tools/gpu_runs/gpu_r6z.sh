#!/bin/bash
# Round-6 measurement pass on the final kernels: probes, full GPU suite + smoke, the bench lines, and rocprofv3
# kernel trace + PMC passes (tools/profile_bench.sh) of every MuJoCo kernel -> profiles/pmc.json
# usage: gpu_r6z.sh [a|b|c|d]   a = probes, suite, smoke, default bench; b = the planar kernels' profiles;
# c = Ant / Pusher / Humanoid profiles (counter groups the roofline needs only); d = bench lines of the families
set -u
export TMPDIR=/tmp
PART=${1:-a}
O=gpurun_out/r6z
mkdir -p $O
if [ "$PART" = a ]; then
bash tools/probe_refs.sh > $O/probe_gpu_box.log 2>&1
( time timeout 2400 python -m pytest tests -m gpu -q ) > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; grep -E "passed|failed|rc=|FAILED|real" $O/gpu_tests.log | tail -8
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $O/gpu_tests.log 2>&1; tail -1 $O/gpu_tests.log
python bench.py > $O/bench_default_before_pmc.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default_before_pmc.json
fi
P() { tag=$1; shift; bash tools/profile_bench.sh $tag "$@" > $O/$tag.log 2>&1; sed -n '/timed window/,/^$/p' gpurun_out/prof_$tag/summary.md | head -3; }
if [ "$PART" = b ] || [ "$PART" = b4 ]; then  # b4: the four lane-group entries only
P r6z_cheetah_lg2
P r6z_cheetah_lg4_8k --num-envs 8192
P r6z_walker_lg2 --task Walker2d
P r6z_hopper_lg1 --task Hopper
fi
if [ "$PART" = b ] || [ "$PART" = b2 ]; then  # the headline kernel at the other two sizes of the throughput table
export PMC_GROUPS=min
P r6z_cheetah_lg2_32k --num-envs 32768
P r6z_cheetah_lg2_128k --num-envs 131072
fi
if [ "$PART" = b ]; then
export PMC_GROUPS=min
P r6z_hopper_lane_f64 --task Hopper --param planar_layout=1
P r6z_cheetah_lane_f64 --param planar_layout=1
P r6z_pusher --task Pusher --num-envs 65536
fi
if [ "$PART" = c ]; then
export PMC_GROUPS=min
P r6z_ant32k_f64 --task Ant --num-envs 32768
P r6z_ant64k_f64 --task Ant --num-envs 65536
P r6z_ant64k_f32 --task Ant --num-envs 65536 --precision fp32
P r6z_humanoid4 --task Humanoid --num-envs 65536
P r6z_standup4 --task HumanoidStandup --num-envs 65536
fi
if [ "$PART" = d ]; then  # after profiles/pmc.json was regenerated from b + c (the lines then carry the VALU roofline)
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
for cfg in "HalfCheetah 8192" "HalfCheetah 32768" "HalfCheetah 131072" "Walker2d 65536" "Hopper 65536" "Ant 32768" "Ant 65536" "Humanoid 65536" "HumanoidStandup 65536" "Pusher 65536"; do
  set -- $cfg
  timeout 600 python bench.py --task $1 --num-envs $2 --no-cpu-baseline 2>>$O/err >> $O/bench.jsonl
done
timeout 300 python bench.py --task Ant --num-envs 65536 --precision fp32 --no-cpu-baseline 2>>$O/err >> $O/bench.jsonl
timeout 300 python bench.py --param planar_layout=1 --no-cpu-baseline 2>>$O/err >> $O/bench.jsonl
timeout 300 python bench.py --task Hopper --param planar_layout=1 --no-cpu-baseline 2>>$O/err >> $O/bench.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r6z/bench.jsonl'):
    d=json.loads(l); print(d['metric'], d['dtype'], d['config']['params'], '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], d['roofline'].get('bound'), 'stale' if d['roofline'].get('stale') else '')
PY
timeout 600 python tools/bench_numpy_api.py > $O/numpy_api.jsonl 2>>$O/err
timeout 900 python tools/bench_families.py > $O/bench_families.md 2>>$O/err
fi
if [ "$PART" = e ]; then  # trip counts of every env on the final build (the one-evaluation search; its soak on the exact search: gpu_r5m.sh)
timeout 900 python tools/lg_iter_soak.py 400 > $O/iter_soak.txt 2>>$O/err; cut -c1-300 $O/iter_soak.txt
fi
