#!/bin/bash
# Profiling recipe for the headline bench (run on the GPU box via gpurun).
# Kernel trace and every PMC group are separate rocprofv3 runs (the guide:
# FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2; never combine --pmc with tracing).
#   usage: tools/profile_bench.sh <tag> [bench args...]
set -u
export TMPDIR=/tmp
TAG=${1:-r1}; shift || true
OUT=gpurun_out/prof_${TAG}
mkdir -p "$OUT"
# one 200-step block of the bench's timed region after 300 warm-up steps (steady state) and nothing else: the reset / numpy-API / async legs launch the
# same kernel on other batch shapes (the async leg: thousands of HALF-size launches) and would pull every
# per-kernel mean -- duration, PMC counts per launch -- towards theirs
ARGS="--steps 200 --warmup 300 --min-time 0 --no-cpu-baseline --only-timed $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python bench.py $ARGS > "$OUT/trace.log" 2>&1
# PMC_GROUPS=min: only what the roofline object needs (HBM bytes, flop instructions) -- the two diagnostic groups
# (wave cycles / waits / instruction classes; VMEM / LDS / L2) are skipped
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE TCC_HIT TCC_MISS"; do
  i=$((i+1))
  if [ "${PMC_GROUPS:-all}" = min ] && { [ $i = 1 ] || [ $i = 5 ]; }; then continue; fi
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc$i" -o p -- python bench.py $ARGS > "$OUT/pmc$i.log" 2>&1
done
python tools/summarize_profile.py "$OUT" > "$OUT/summary.md" 2>&1
# raw per-dispatch tables are large (gpurun merges at most 64 MiB back): keep the summaries
find "$OUT" -name '*counter_collection.csv' -delete
find "$OUT" -name '*kernel_trace.csv' -delete
cat "$OUT/summary.md"
