#!/bin/bash
T=gpurun_out/r04f; mkdir -p $T
python tools/sk_table.py /tmp/sk_convdet.json W:768:72:37440
for v in "5 1338" "5 1300" "10 1338" "4 1000"; do
  set -- $v
  SQD_TUNING_JSON=/tmp/sk_convdet.json SQD_SK_KSPLIT=$1 SQD_SK_HBIAS=$2 timeout -k 10 200 python bench.py --mode infer --layers --no-cpu-baseline --no-pipeline > $T/infer_ks$1_hb$2.json 2> $T/infer_ks$1_hb$2.err; echo "ks $1 hb $2 rc $?"
done
timeout -k 10 200 python bench.py --mode infer --layers --no-cpu-baseline --no-pipeline > $T/infer_base.json 2> $T/infer_base.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04f/infer_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    L = (d.get('layers') or {}).get('infer') or {}
    print(f.split('/')[-1], 'ms', d.get('ms_per_step'), {k.split('|')[0].strip(): v for k, v in L.items() if 'C768 N72' in k})
PY
timeout -k 10 300 python -m pytest tests/test_padcrop.py tests/test_headline_gpu.py tests/test_dropout_gpu.py tests/test_surface_gpu.py -q -m gpu -x -k "padcrop or float64 or shift or forbid or dropout or launch_plan" > $T/pytest_new.log 2>&1; echo "pytest rc $?"; tail -5 $T/pytest_new.log
