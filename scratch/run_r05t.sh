#!/bin/bash
# the tables scratch/tuning_gw_<v>.json = the shipped tuning.json + one row 'GW:<tc>:<blocks>:<npix>': {'cfg': S} each (a: GW:2:52:37440=19, b: GW:1:18:37440=14,
# c: =56, d: GW:2:4:149760=64, e: =256, f: GW:1:2:599040=128, g: =512, h: GW:2:52:37440=4); results in DESIGN.md section 3
O=gpurun_out/r05t; mkdir -p $O
run() {
  timeout -k 10 200 python bench.py --mode train --steps 40 --warmup 10 --no-cpu-baseline --layers > $O/train_$1.json 2> $O/train_$1.err
  python - $O/train_$1.json $1 <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
L = d['layers']['train']
print(sys.argv[2], 'ms/step', d['ms_per_step'], {k.split('|')[1].strip()[11:]: v for k, v in L.items() if 'wgrad_wino_group' in k}, 'reduce', [v for k, v in L.items() if 'reduce_batched' in k])
PY
}
run base
for v in a b c d e f g h; do SQD_TUNING_JSON=$PWD/scratch/tuning_gw_$v.json run $v; done
run base2
