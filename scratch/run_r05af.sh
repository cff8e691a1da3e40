#!/bin/bash
# the tables scratch/tuning_g1_<v>.json = tuning.json + one row 'G1:<tc>:<tile groups>:<npix>': {'cfg': S} (a-d: the fire11-14 expand1x1 group at
# S = 8 / 12 / 24 / 32 instead of 16; e, f: the fire9 / 10 group at 42 / 170 instead of 85)
O=gpurun_out/r05af; mkdir -p $O
run() {
  timeout -k 10 200 python bench.py --mode train --steps 40 --warmup 10 --no-cpu-baseline --layers > $O/train_$1.json 2> $O/train_$1.err
  python - $O/train_$1.json $1 <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
L = d['layers']['train']
print(sys.argv[2], 'ms/step', d['ms_per_step'], {k.split('|')[1].strip()[11:]: v for k, v in L.items() if 'conv_wgrad_group<1>' in k}, 'reduce', [v for k, v in L.items() if 'reduce_batched' in k])
PY
}
run base
for v in a b c d e f; do SQD_TUNING_JSON=$PWD/scratch/tuning_g1_$v.json run $v; done
run base2
