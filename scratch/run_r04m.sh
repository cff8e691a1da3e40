#!/bin/bash
T=gpurun_out/r04m; mkdir -p $T
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "storing_form or training_form or fire_bridge_one_launch or stem_pool" > $T/pytest_new.log 2>&1; echo "pytest new rc $?"; tail -5 $T/pytest_new.log
timeout -k 10 900 python -m pytest tests/test_training_gpu.py tests/test_headline_gpu.py -x -q --deselect tests/test_training_gpu.py::test_training_launch_plan_equals_real_launches > $T/pytest_train.log 2>&1; echo "pytest train rc $?"; tail -5 $T/pytest_train.log
for v in 1 0; do
SQD_FUSE_TRAIN_FWD=$v timeout -k 10 300 python bench.py --mode train --no-cpu-baseline --no-pipeline --layers > $T/train_f$v.json 2> $T/train_f$v.err; echo "bench f$v rc $?"
done
python - <<'PY'
import json
for v in (1,0):
    d=json.loads(open(f'gpurun_out/r04m/train_f{v}.json').read().strip().splitlines()[-1])
    t=d.get('train') or d
    print('fuse',v,'ms',d['ms_per_step'],d.get('repeat_window_ms_per_step'))
    L=d['layers']['train']
    for k,x in L.items():
        if '96x312' in k or '384x1248' in k: print('   ',k,x)
PY
