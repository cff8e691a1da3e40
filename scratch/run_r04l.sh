#!/bin/bash
T=gpurun_out/r04l; mkdir -p $T
timeout -k 10 500 python bench.py > $T/bench_default.json 2> $T/bench_default.err; echo "default rc $?"
timeout -k 10 300 python bench.py --mode infer --inflight 1 --no-cpu-baseline > $T/bench_inflight1.json 2> $T/bench_inflight1.err; echo "inflight1 rc $?"
timeout -k 10 300 python bench.py --mode infer --inflight 3 --no-cpu-baseline --no-pipeline > $T/bench_inflight3.json 2> $T/bench_inflight3.err; echo "inflight3 rc $?"
python - <<'PY'
import json
for f in ('bench_default','bench_inflight1','bench_inflight3'):
    try:
        d=json.loads(open(f'gpurun_out/r04l/{f}.json').read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    print(f, d['value'], d['ms_per_step'], d.get('steps_in_flight'), d.get('serial_ms_per_step'), d.get('timed_with'))
    print('   roofline', d.get('roofline'))
    p=d.get('pipeline') or d.get('end_to_end')
    print('   pipeline', p)
    t=d.get('training')
    if t: print('   training', {k:t[k] for k in t if k in ('value','ms_per_step','images_per_s')})
PY
tail -5 $T/bench_default.err
