#!/bin/bash
T=gpurun_out/r04w; mkdir -p $T
timeout -k 10 400 python bench.py --force-dist --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline > $T/bench_forcedist.json 2> $T/bench_forcedist.err; echo "forcedist rc $?"; tail -3 $T/bench_forcedist.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04w/bench_forcedist.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','n_gpus','ms_per_step','timed_with','steps_in_flight','serial_ms_per_step') if k in d})
print({k:d['train'][k] for k in ('value','ms_per_step','timed_with') if k in d['train']})
PY
timeout -k 10 600 python -m pytest tests/test_data_parallel_gpu.py tests/test_surface_gpu.py -x -q > $T/pytest_dp.log 2>&1; echo "pytest dp rc $?"; tail -3 $T/pytest_dp.log
