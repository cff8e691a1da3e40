#!/bin/bash
# grouped Winograd weight-gradient launch: parity, then A/B inside the training step
mkdir -p gpurun_out/r05o
O=gpurun_out/r05o
timeout -k 10 300 python -m pytest tests/test_wgrad_group_gpu.py tests/test_training_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
for g in 1 0 1 0; do
  SQD_WW_GROUP=$g timeout -k 10 200 python bench.py --mode train --steps 40 --warmup 10 --no-cpu-baseline --layers > $O/train_g$g.json 2> $O/train_g$g.err
  python - $O/train_g$g.json $g <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
k = d.get('kernels_event_profile', {})
print('group', sys.argv[2], 'ms/step', d['ms_per_step'], 'img/s', d['value'],
      {n: round(v['ms_per_step'], 4) for n, v in k.items() if 'wgrad' in n or 'squeeze_bwd' in n})
PY
done
