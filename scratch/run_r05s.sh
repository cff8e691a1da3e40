#!/bin/bash
O=gpurun_out/r05s; mkdir -p $O
echo "--- round-4 input kernels (isolated)"; PRE_LIB=scratch/libpre_r04.so timeout -k 10 100 python scratch/pre_time.py 2>&1 | tail -2
echo "--- this tree"; timeout -k 10 100 python scratch/pre_time.py 2>&1 | tail -2
timeout -k 10 200 python -m pytest tests/test_preprocess.py tests/test_padcrop.py tests/test_lanes_gpu.py -x -q -m gpu 2>&1 | tail -1
for k in 1 2; do
timeout -k 10 300 python bench.py --mode infer --no-cpu-baseline > $O/bench$k.json 2> $O/bench$k.err; echo "bench rc $?"
python - $O/bench$k.json <<'PY'
import json, sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('value', d['value'], d['ms_per_step'], 'pipeline', d['pipeline']['value'], 'dataset', d['detect_dataset']['value'], d['detect_dataset'].get('ms_per_batch'), d['detect_dataset'].get('loader_threads'))
PY
done
