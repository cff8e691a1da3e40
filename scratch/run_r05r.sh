#!/bin/bash
O=gpurun_out/r05r; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_preprocess.py tests/test_padcrop.py tests/test_lanes_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout -k 10 100 python scratch/pre_time.py 2>&1 | tail -3
timeout -k 10 300 python bench.py --mode infer --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05r/bench.json') if l.startswith('{')][-1])
print('value', d['value'], d['ms_per_step'], 'pipeline', d['pipeline']['value'], 'dataset', d['detect_dataset']['value'], d['detect_dataset'].get('ms_per_batch'))
PY
