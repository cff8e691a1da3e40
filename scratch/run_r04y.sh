#!/bin/bash
T=gpurun_out/r04y; mkdir -p $T
for z in 0 2 3 6 8 12; do
  if [ $z = 0 ]; then timeout -k 10 300 python bench.py --mode train --no-cpu-baseline --no-pipeline --layers > $T/train_z$z.json 2> $T/train_z$z.err
  else SQD_ZSEG_TRAIN=$z timeout -k 10 300 python bench.py --mode train --no-cpu-baseline --no-pipeline --layers > $T/train_z$z.json 2> $T/train_z$z.err; fi
  python - <<PY
import json
d=json.loads(open("$T/train_z$z.json").read().strip().splitlines()[-1])
L=d['layers']['train']
print("zseg $z", 'ms', d['ms_per_step'], d.get('repeat_window_ms_per_step'), [v for k,v in L.items() if 'fire_pool_bridge_save' in k])
PY
done
