#!/bin/bash
T=gpurun_out/r04zz_final; mkdir -p $T
timeout -k 10 900 python -m pytest tests -q -m gpu > $T/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -2 $T/pytest_gpu.log
timeout -k 10 400 python bench.py > $T/bench_default.json 2> $T/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04zz_final/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('serial_ms_per_step'), d['roofline']['frac'], d['train']['value'], d['train']['ms_per_step'], d['parity']['ok'])
PY
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $T/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $T/smoke.log
