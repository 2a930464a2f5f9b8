#!/bin/bash
# Round 6, second session: host placement (helper threads on the device's NUMA node; bind_host_to_device) A/B
export TMPDIR=/tmp
O=gpurun_out/r6y; mkdir -p $O
{
python -c "import envpool_amd as e; print('device node', e.device_numa_node(0), 'cpus', len(e.device_local_cpus(0)))"
for rep in 1 2 3; do
  for mode in far near none bind; do
    case $mode in
      far)  pre="taskset -c 0-63"; arg="" ;;
      near) pre="taskset -c 64-127"; arg="" ;;
      none) pre=""; arg="" ;;
      bind) pre=""; arg="bind" ;;
    esac
    echo "mode=$mode rep=$rep $($pre python tools/numpy_step_ab.py HalfCheetah 65536 32768 6 $arg 2>/dev/null | tail -1)"
  done
done
} 2>&1 | tee $O/numa_ab.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6y/bench_default.json').read().strip().splitlines()[-1])
print('value %.4e' % d['value'], 'numpy_api %.3e' % d['numpy_api']['value'], 'async %.3e' % d['async_mode']['value'], d['config']['host_binding'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('openmp_port',{}).get('value'))
PY
python bench.py --no-bind --no-cpu-baseline > $O/bench_nobind.json 2>> $O/bench_default.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6y/bench_nobind.json').read().strip().splitlines()[-1])
print('nobind value %.4e' % d['value'], 'numpy_api %.3e' % d['numpy_api']['value'], 'async %.3e' % d['async_mode']['value'], d['config']['host_binding'])
PY
