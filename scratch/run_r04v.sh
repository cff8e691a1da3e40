#!/bin/bash
T=gpurun_out/r04v; mkdir -p $T
timeout -k 10 500 python bench.py --gpus 2 --backend gloo --steps 10 --warmup 3 --no-cpu-baseline --no-pipeline > $T/bench_gloo2.json 2> $T/bench_gloo2.err; echo "gloo2 rc $?"; tail -3 $T/bench_gloo2.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r04v/bench_gloo2.json').read().strip().splitlines()[-1])
    print({k:d[k] for k in ('value','n_gpus','ms_per_step','scaling','timed_with','steps_in_flight','serial_ms_per_step') if k in d})
    print({k:d['train'][k] for k in ('value','ms_per_step','timed_with') if k in d['train']})
    print(d['config'])
except Exception as e: print('unreadable', e)
PY
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --steps 10 --warmup 3 --no-cpu-baseline --no-pipeline --mode infer > $T/bench_torchrun2.json 2> $T/bench_torchrun2.err; echo "torchrun2 rc $?"; tail -2 $T/bench_torchrun2.json | cut -c1-400
