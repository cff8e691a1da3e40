#!/bin/bash
# the table scratch/tuning_c48_<k>.json = tuning.json with row 'W:48:192:37440' (fire9 / fire10 expand3x3) set to the U-stationary configuration k
O=gpurun_out/r05ad; mkdir -p $O
run() {
  timeout -k 10 300 python bench.py --mode infer --no-cpu-baseline --no-pipeline --layers > $O/bench_$1.json 2> $O/bench_$1.err
  python - $O/bench_$1.json $1 <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
L = d['layers']['infer'] if 'infer' in d['layers'] else d['layers']
print(sys.argv[2], 'value', d['value'], d['ms_per_step'], 'serial', d['serial_ms_per_step'], 'frac', d['roofline']['frac'], {k: v for k, v in L.items() if 'C48 N192' in k})
PY
}
run base
for v in 10 8; do SQD_TUNING_JSON=$PWD/scratch/tuning_c48_$v.json run c$v; done
run base2
