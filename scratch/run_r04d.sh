#!/bin/bash
T=gpurun_out/r04d; mkdir -p $T
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $T/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -15 $T/pytest_gpu.log
python tools/sk_table.py /tmp/sk_cd.json W:768:72:37440 W:72:768:37440
for tag in base sk_cd; do
  if [ $tag = base ]; then unset SQD_TUNING_JSON; else export SQD_TUNING_JSON=/tmp/$tag.json; fi
  timeout -k 10 300 python bench.py --layers --no-cpu-baseline --no-pipeline > $T/bench_$tag.json 2> $T/bench_$tag.err; echo "bench $tag rc $?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04d/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    t = d.get('train') or {}
    print(f.split('/')[-1], 'infer ms', d.get('ms_per_step'), 'train ms', t.get('ms_per_step'), 'parity', d.get('parity', {}).get('ok'))
    for m, L in (d.get('layers') or {}).items():
        for k, v in (L or {}).items():
            if ('sk' in k or 'C768 N72' in k or 'C72 N768' in k or 'dropout' in k or 'sumsq' in k): print('   ', m, k, v)
PY
