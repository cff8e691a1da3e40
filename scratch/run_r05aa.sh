#!/bin/bash
O=gpurun_out/r05aa; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_wgrad_group_gpu.py tests/test_training_gpu.py tests/test_surface_gpu.py tests/test_headline_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
for k in 1 2; do
timeout -k 10 200 python bench.py --mode train --steps 40 --warmup 10 --no-cpu-baseline --layers > $O/train$k.json 2> $O/train$k.err
python - $O/train$k.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
L = d['layers']['train']
print('ms/step', d['ms_per_step'], {k: v for k, v in L.items() if 'wgrad' in k and '1tap' in k}, 'reduce', [v for k, v in L.items() if 'reduce_batched' in k])
PY
done
