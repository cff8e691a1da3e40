#!/bin/bash
T=gpurun_out/r04e; mkdir -p $T
python tools/sk_table.py /tmp/sk_convdet.json W:768:72:37440
export SQD_TUNING_JSON=/tmp/sk_convdet.json
for ks in 1 2 3 4 6 8 0; do
  SQD_SK_KSPLIT=$ks timeout -k 10 200 python bench.py --mode infer --layers --no-cpu-baseline --no-pipeline > $T/infer_ks$ks.json 2> $T/infer_ks$ks.err; echo "ks $ks rc $?"
done
unset SQD_TUNING_JSON
timeout -k 10 200 python bench.py --mode infer --layers --no-cpu-baseline --no-pipeline > $T/infer_base.json 2> $T/infer_base.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04e/infer_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    L = (d.get('layers') or {}).get('infer') or {}
    print(f.split('/')[-1], 'ms', d.get('ms_per_step'), {k.split('|')[0].strip(): v for k, v in L.items() if 'C768 N72' in k})
PY
timeout -k 10 300 python -m pytest tests/test_padcrop.py tests/test_headline_gpu.py -q -m gpu -x -k "padcrop or float64 or shift or forbid" > $T/pytest_new.log 2>&1; echo "pytest rc $?"; tail -5 $T/pytest_new.log
