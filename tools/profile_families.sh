#!/bin/bash
# rocprofv3 evidence for the HBM-streaming kernels (K1 classic_control, K2 toy_text, K4 Atari post-process):
# kernel trace, FETCH_SIZE and WRITE_SIZE as three separate runs of the same tools/bench_families.py command
# (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2; never --pmc together with tracing).
#   usage: tools/profile_families.sh <tag> [bench_families args...]
set -u
export TMPDIR=/tmp
TAG=${1:-r5}; shift || true
OUT=gpurun_out/prof_${TAG}
mkdir -p "$OUT"
ARGS="--steps 200 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python tools/bench_families.py $ARGS --plan-out $OUT/plan.json > "$OUT/trace.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -o p -- python tools/bench_families.py $ARGS --plan-out $OUT/plan_pmc_$c.json > "$OUT/pmc_$c.log" 2>&1
done
python tools/summarize_families.py "$OUT" --json "$OUT/families_pmc.json" > "$OUT/summary.md" 2>&1
find "$OUT" -name '*counter_collection.csv' -delete
find "$OUT" -name '*kernel_trace.csv' -delete
cat "$OUT/summary.md"
