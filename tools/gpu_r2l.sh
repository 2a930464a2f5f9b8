#!/bin/bash
# PMC passes for the quad Humanoid kernel: where do the wave cycles go?
set -u
export TMPDIR=/tmp
O=gpurun_out/r2l
mkdir -p $O
ARGS="--task Humanoid --num-envs 65536 --steps 30 --warmup 10 --no-cpu-baseline"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d "$O/pmc$i" -o p -- python bench.py $ARGS > "$O/pmc$i.log" 2>&1
done
python - <<'PY' > $O/summary.txt
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob('gpurun_out/r2l/pmc*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'Humanoid4StepKernel' in r.get('Kernel_Name',''):
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    v=agg[k][5:] if len(agg[k])>10 else agg[k]
    print(k, len(v), sum(v)/len(v))
PY
find $O -name '*counter_collection.csv' -delete
cat $O/summary.txt
tail -3 $O/pmc3.log
