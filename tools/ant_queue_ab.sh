#!/bin/bash
# A/B of the Ant kernel's work queue (mujoco_ant.hip): mj_steps per unit ("ant_sub": 5 = a chunk's whole env-step
# is one unit, the schedule of rounds 2-5; 1 = substep-granular) x cost-sorted rows ("ant_sort").  Interleaved,
# REPS passes; one bench line per run -> $OUT (profiles/r6*_ant_queue_ab.jsonl).
N=${N:-32768}; REPS=${REPS:-2}; OUT=${OUT:-gpurun_out/ant_queue_ab.jsonl}; PREC=${PREC:-fp64}
: > "$OUT"
for rep in $(seq $REPS); do
  for cfg in "5 0" "5 1" "1 0" "1 1" ${EXTRA}; do
    set -- $cfg
    line=$(python bench.py --task Ant --num-envs $N --precision $PREC --only-timed --no-cpu-baseline --min-time 3 \
           --param ant_sub=$1 --param ant_sort=$2 2>/dev/null | tail -1)
    echo "{\"ant_sub\": $1, \"ant_sort\": $2, \"rep\": $rep, \"line\": $line}" >> "$OUT"
  done
done
python - "$OUT" <<'PY'
import json, sys, collections
rows = [json.loads(l) for l in open(sys.argv[1])]
agg = collections.defaultdict(list)
for r in rows:
    agg[(r["ant_sub"], r["ant_sort"])].append((r["line"]["value"], r["line"].get("kernel_ms_per_launch") or r["line"].get("ms_per_step")))
print("| ant_sub | ant_sort | env-steps/s (runs) | ms per step |")
print("|---|---|---|---|")
for k, v in sorted(agg.items()):
    print(f"| {k[0]} | {k[1]} | " + " / ".join(f"{x[0]:.3e}" for x in v) + " | " + " / ".join(f"{x[1]:.4f}" for x in v) + " |")
PY
