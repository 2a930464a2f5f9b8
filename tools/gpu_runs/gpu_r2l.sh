#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2l
rm -rf $O; mkdir -p $O
ARGS="--task Humanoid --num-envs 65536 --steps 30 --warmup 10 --no-cpu-baseline"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
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
    v=agg[k][10:] if len(agg[k])>20 else agg[k]
    print(k, len(v), "%.4g"%(sum(v)/len(v)))
f=sum(agg['FETCH_SIZE'][10:])/len(agg['FETCH_SIZE'][10:]); w=sum(agg['WRITE_SIZE'][10:])/len(agg['WRITE_SIZE'][10:])
print("traffic GB per launch", 1024*(2*f+w)/1e9)
PY
find $O -name '*counter_collection.csv' -delete
cat $O/summary.txt
