#!/bin/bash
# the tables scratch/tuning_w<k>.json = the shipped tuning.json with row 'W:64:16:599040' set to Winograd configuration k (9 = U-stationary <1,8>,
# 10 = U-stationary <2,4>): 255 / 281 us for the two fire3 / fire4 data gradients against 224 for the shipped <1,4>
O=gpurun_out/r05x; mkdir -p $O
run() {
  timeout -k 10 200 python bench.py --mode train --steps 40 --warmup 10 --no-cpu-baseline --layers > $O/train_$1.json 2> $O/train_$1.err
  python - $O/train_$1.json $1 <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
L = d['layers']['train']
print(sys.argv[2], 'ms/step', d['ms_per_step'], {k: v for k, v in L.items() if '9tap C64 N16' in k or '9tap C128 N32' in k})
PY
}
run base
for v in w9 w8 w10; do SQD_TUNING_JSON=$PWD/scratch/tuning_$v.json run $v; done
