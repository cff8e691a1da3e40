#!/bin/bash
# round 5, first GPU call: the lane executor's tests, the dataset driver, the default bench line through the package API
T=gpurun_out/r05a; mkdir -p $T
timeout -k 10 500 python -m pytest tests/test_lanes_gpu.py tests/test_preprocess.py tests/test_conv_wino_sk_gpu.py -x -q -m gpu > $T/pytest_lanes.log 2>&1; echo "pytest rc $?"; tail -15 $T/pytest_lanes.log
timeout -k 10 400 python bench.py > $T/bench_default.json 2> $T/bench_default.err; echo "bench rc $?"; tail -5 $T/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05a/bench_default.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'serial', d.get('serial_ms_per_step'), 'degraded', d.get('degraded'), 'frac', d['roofline']['frac'])
print('timed_with', d['timed_with'])
print('train', d['train']['value'], d['train']['ms_per_step'], 'parity', d['parity']['ok'])
print('pipeline', d.get('pipeline', {}).get('value'), d.get('pipeline', {}).get('ms_per_step'), d.get('pipeline', {}).get('degraded'))
print('detect_dataset', d.get('detect_dataset'))
PY
