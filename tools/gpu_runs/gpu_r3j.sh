#!/bin/bash
# round 3: mid-round check on the current build -- full GPU suite + smoke, default bench lines,
# per-family table with the compiled reference's CPU rate beside it, numpy-API async lines, and the
# 8-rank plumbing dry run of BASELINE config 4 (8 ranks sharing ONE GPU over gloo: not a scaling number)
set -u
export TMPDIR=/tmp
O=gpurun_out/r3j
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s --durations=10 > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; grep -E "passed|failed|rc=|FAILED" $O/gpu_tests.log | tail -8
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $O/gpu_tests.log 2>&1; tail -1 $O/gpu_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; cut -c1-330 $O/bench_default.json
for t in Walker2d Hopper Ant Pusher; do timeout 300 python bench.py --task $t --no-cpu-baseline 2>>$O/err | tee -a $O/bench.jsonl | cut -c1-200; done
timeout 300 python bench.py --num-envs 8192 --no-cpu-baseline 2>>$O/err | tee -a $O/bench.jsonl | cut -c1-200
timeout 900 python tools/bench_families.py --steps 100 > $O/bench_families.md 2>>$O/err; tail -34 $O/bench_families.md
timeout 600 python tools/bench_reference_cpu.py > $O/reference_cpu.jsonl 2>>$O/err; cat $O/reference_cpu.jsonl
timeout 600 python tools/bench_async_api.py > $O/async_numpy_api.jsonl 2>>$O/err; cat $O/async_numpy_api.jsonl
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 8 --backend gloo --allgather --task Ant --num-envs 32768 --steps 20 --warmup 5 > $O/dryrun_8ranks_one_gpu.json 2> $O/dryrun.err; cut -c1-400 $O/dryrun_8ranks_one_gpu.json; tail -2 $O/dryrun.err
