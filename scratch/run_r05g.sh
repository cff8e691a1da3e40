#!/bin/bash
T=gpurun_out/r05g; mkdir -p $T
timeout -k 10 600 python -m pytest tests/test_headline_gpu.py tests/test_inference_gpu.py -x -q -m gpu > $T/pytest_headline.log 2>&1; echo "pytest rc $?"; tail -4 $T/pytest_headline.log
timeout -k 10 400 python bench.py --layers > $T/bench_layers.json 2> $T/bench_layers.err; echo "bench rc $?"; tail -3 $T/bench_layers.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05g/bench_layers.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'serial', d.get('serial_ms_per_step'), 'degraded', d.get('degraded'), 'roofline', d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
print('train', d['train']['value'], d['train']['ms_per_step'], 'parity', d['parity'])
for k,v in sorted(d['layers']['infer'].items(), key=lambda kv:-kv[1])[:14]: print('  infer', k, v)
for k,v in sorted(d['layers']['train'].items(), key=lambda kv:-kv[1])[:14]: print('  train', k, v)
print('pipeline', d['pipeline']['value'], 'dataset', d['detect_dataset']['value'])
PY
