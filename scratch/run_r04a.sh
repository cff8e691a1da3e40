#!/bin/bash
# round 4, first GPU call: parity of the balanced Winograd kernel, isolated and in-step timing against the unit kernel
T=gpurun_out/r04a; mkdir -p $T
timeout -k 10 400 python -m pytest tests/test_conv_wino_sk_gpu.py -x -q > $T/pytest_sk.log 2>&1; rc=$?; echo "pytest sk rc $rc"; tail -5 $T/pytest_sk.log
[ $rc -ne 0 ] && exit 1
WCFGS=2,16 ITERS=30 timeout -k 10 200 python scratch/wino_cfgs_bench.py > $T/isolated.log 2>&1; echo "isolated rc $?"; cat $T/isolated.log
F="W:768:72:37440 W:96:384:37440 W:48:192:37440"
python tools/sk_table.py /tmp/sk_convdet.json W:768:72:37440
python tools/sk_table.py /tmp/sk_fwd.json $F
python tools/sk_table.py /tmp/sk_fwd4.json $F W:64:256:37440
python tools/sk_table.py /tmp/sk_all.json $F W:72:768:37440 W:384:96:37440 W:192:48:37440 W:256:64:37440
for tag in base sk_convdet sk_fwd sk_fwd4; do
  if [ $tag = base ]; then unset SQD_TUNING_JSON; else export SQD_TUNING_JSON=/tmp/$tag.json; fi
  timeout -k 10 200 python bench.py --mode infer --layers --no-cpu-baseline --no-pipeline > $T/bench_infer_$tag.json 2> $T/bench_infer_$tag.err; echo "bench infer $tag rc $?"
done
for tag in base sk_fwd sk_all; do
  if [ $tag = base ]; then unset SQD_TUNING_JSON; else export SQD_TUNING_JSON=/tmp/$tag.json; fi
  timeout -k 10 240 python bench.py --mode train --layers --no-cpu-baseline --no-pipeline > $T/bench_train_$tag.json 2> $T/bench_train_$tag.err; echo "bench train $tag rc $?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04a/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    t = d.get('train')
    print(f.split('/')[-1], 'infer ms', d.get('ms_per_step') if not t or 'value' in d else None, 'train ms', t.get('ms_per_step') if t else None)
    lay = (d.get('layers') or {})
    for m, L in lay.items():
        if not L: continue
        for k, v in L.items():
            if 'wino' in k and ('24x78' in k): print('   ', m, k, v)
PY
export TMPDIR=/tmp; (cd /tmp && rocprofv3 -L > $OLDPWD/$T/counters.txt 2>&1); echo "counters rc $?"; grep -c . $T/counters.txt
