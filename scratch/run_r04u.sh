#!/bin/bash
T=gpurun_out/r04u; mkdir -p $T
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_training_gpu.py -x -q -k "wgrad" > $T/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $T/pytest.log
for v in "base" "SQD_WW_TARGET41=512" "SQD_WW_TC1=1" "SQD_WW_TC1=1 SQD_WW_TARGET41=1024"; do
  tag=$(echo $v | tr '= ' '__')
  if [ "$v" = base ]; then timeout -k 10 300 python bench.py --mode train --no-cpu-baseline --no-pipeline --layers > $T/train_$tag.json 2> $T/train_$tag.err
  else env $v timeout -k 10 300 python bench.py --mode train --no-cpu-baseline --no-pipeline --layers > $T/train_$tag.json 2> $T/train_$tag.err; fi
  python - <<PY
import json
d=json.loads(open("$T/train_$tag.json").read().strip().splitlines()[-1])
k=d['kernels_event_profile']
L=d['layers']['train']
print("$v", 'ms', d['ms_per_step'], d.get('repeat_window_ms_per_step'), 'wgrad_wino', round(k['conv_wgrad_wino']['ms_per_step']*1e3,1), 'reduce', round(k['wgrad_reduce_batched']['ms_per_step']*1e3,1))
print('    ', {kk.split('9tap ')[1]: v for kk,v in L.items() if 'wgrad_wino' in kk})
PY
done
