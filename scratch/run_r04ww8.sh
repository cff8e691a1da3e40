#!/bin/bash
T=gpurun_out/r04ww8; mkdir -p $T
SQD_WW8=1 timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_training_gpu.py -x -q -k "wgrad" > $T/pytest_ww8.log 2>&1; echo "pytest ww8 rc $?"; tail -3 $T/pytest_ww8.log
for v in "SQD_WW8=0" "SQD_WW8=t1" "SQD_WW8=1" "SQD_WW8=1 SQD_WW_TC1=1"; do
  tag=$(echo $v | tr '= ' '__')
  env $v timeout -k 10 300 python bench.py --mode train --no-cpu-baseline --no-pipeline --layers > $T/train_$tag.json 2> $T/train_$tag.err
  python - <<PY
import json
d=json.loads(open("$T/train_$tag.json").read().strip().splitlines()[-1])
k=d['kernels_event_profile']
L=d['layers']['train']
print("$v", 'ms', d['ms_per_step'], d.get('repeat_window_ms_per_step'), 'wgrad_wino', round(k['conv_wgrad_wino']['ms_per_step']*1e3,1))
print('    ', {kk.split('9tap ')[1]: v for kk,v in L.items() if 'wgrad_wino' in kk})
PY
done
