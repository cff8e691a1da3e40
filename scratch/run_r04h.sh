#!/bin/bash
T=gpurun_out/r04h; mkdir -p $T
timeout -k 10 600 python -m pytest tests/test_inference_gpu.py tests/test_headline_gpu.py tests/test_preprocess.py tests/test_padcrop.py -q -m gpu -x > $T/pytest_det.log 2>&1; echo "pytest rc $?"; tail -4 $T/pytest_det.log
timeout -k 10 300 python bench.py --layers --no-cpu-baseline > $T/bench_a.json 2> $T/bench_a.err; echo "bench rc $?"
SQD_WW52=1 timeout -k 10 300 python bench.py --mode train --layers --no-cpu-baseline --no-pipeline > $T/bench_ww52.json 2> $T/bench_ww52.err; echo "bench ww52 rc $?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04h/bench_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    t = d.get('train') or (d if 'train' not in d and d.get('config',{}).get('workload','').find('training')>=0 else {})
    print(f.split('/')[-1], 'infer ms', d.get('ms_per_step'), 'train ms', (d.get('train') or {}).get('ms_per_step'), 'pipeline', (d.get('pipeline') or {}).get('value'))
    for m, L in (d.get('layers') or {}).items():
        for k, v in (L or {}).items():
            if ('detect' in k or 'wgrad 9tap C768' in k): print('   ', m, k, v)
PY
