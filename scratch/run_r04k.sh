#!/bin/bash
T=gpurun_out/r04k; mkdir -p $T
for v in "base" "SQD_WW_TC1=1" "SQD_WW_TARGET=384" "SQD_WW_TARGET=768" "SQD_WW52=1"; do
  tag=$(echo $v | tr '=' '_')
  if [ "$v" = base ]; then timeout -k 10 300 python bench.py --mode train --no-cpu-baseline --no-pipeline > $T/train_$tag.json 2> $T/train_$tag.err
  else env $v timeout -k 10 300 python bench.py --mode train --no-cpu-baseline --no-pipeline > $T/train_$tag.json 2> $T/train_$tag.err; fi
  python - <<PY
import json
d=json.loads(open("$T/train_$tag.json").read().strip().splitlines()[-1])
k=d['kernels_event_profile']
print("$v", 'ms', d['ms_per_step'], d.get('repeat_window_ms_per_step'), 'wgrad_wino', k['conv_wgrad_wino']['ms_per_step'], 'reduce', k['wgrad_reduce_batched']['ms_per_step'])
PY
done
