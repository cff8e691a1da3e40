#!/bin/bash
T=gpurun_out/r05l; mkdir -p $T
for tab in shipped vp; do
  if [ $tab = vp ]; then export SQD_TUNING_JSON=$PWD/scratch/tuning_vp.json; else unset SQD_TUNING_JSON; fi
  timeout -k 10 300 python bench.py --mode infer --no-cpu-baseline --no-pipeline --layers > $T/bench_$tab.json 2> $T/bench_$tab.err; echo "bench $tab rc $?"
done
python - <<'PY'
import json
for tab in ('shipped','vp'):
    d=json.loads(open(f'gpurun_out/r05l/bench_{tab}.json').read().strip().splitlines()[-1])
    print(tab, 'value', d['value'], 'ms', d['ms_per_step'], 'serial', d['serial_ms_per_step'])
    for k,v in sorted(d['layers']['infer'].items(), key=lambda kv:-kv[1])[:4]: print('   ', k, v)
PY
