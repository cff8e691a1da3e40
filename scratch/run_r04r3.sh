#!/bin/bash
T=gpurun_out/r04r3; mkdir -p $T
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_training_gpu.py -x -q -k "wgrad or backward" > $T/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $T/pytest.log
timeout -k 10 300 python bench.py --mode train --no-cpu-baseline --no-pipeline --layers > $T/train.json 2> $T/train.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04r3/train.json').read().strip().splitlines()[-1])
print('ms',d['ms_per_step'],d.get('repeat_window_ms_per_step'))
L=d['layers']['train']
for k,x in L.items():
    if 'wgrad_wino' in k: print('   ',k,x)
PY
timeout -k 10 600 python tools/pmc_per_launch.py r04r3 train "FETCH_SIZE" > $T/pmc.log 2>&1; echo "pmc rc $?"
grep "wgrad_wino" gpurun_out/pmc_launch_r04r3_train.txt
