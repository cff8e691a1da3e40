#!/bin/bash
T=gpurun_out/r05d; mkdir -p $T
timeout -k 10 300 python -m pytest tests/test_lanes_gpu.py -x -q -m gpu > $T/pytest_lanes.log 2>&1; echo "pytest rc $?"; tail -3 $T/pytest_lanes.log
for L in 2 3; do
timeout -k 10 300 python bench.py --mode infer --no-cpu-baseline --inflight $L > $T/bench_infer_L$L.json 2> $T/bench_infer_L$L.err; echo "bench rc $?"; tail -3 $T/bench_infer_L$L.err
python - $L <<'PY'
import json,sys
L=sys.argv[1]
d=json.loads(open(f'gpurun_out/r05d/bench_infer_L{L}.json').read().strip().splitlines()[-1])
print('L',L,'value', d['value'], 'ms', d['ms_per_step'], 'serial', d.get('serial_ms_per_step'), 'degraded', d.get('degraded'))
print('pipeline', d.get('pipeline', {}).get('value'), d.get('pipeline', {}).get('ms_per_step'))
dd=d.get('detect_dataset') or {}
print('detect_dataset', dd.get('value'), dd.get('ms_per_batch'))
PY
done
