#!/bin/bash
# A/B: conv_wino_kernel's output stores with the non-temporal hint (scratch/libsqdhip_nt.so = the tree built with -DSQD_WINO_NT_STORE)
O=gpurun_out/r05y; mkdir -p $O
run() {
  timeout -k 10 300 python bench.py --mode infer --no-cpu-baseline --no-pipeline --layers > $O/bench_$1.json 2> $O/bench_$1.err
  python - $O/bench_$1.json $1 <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
L = d['layers']['infer'] if 'infer' in d['layers'] else d['layers']
print(sys.argv[2], 'value', d['value'], d['ms_per_step'], 'serial', d['serial_ms_per_step'], {k: v for k, v in L.items() if 'conv_wino<2,4>' in k or 'C384' in k or 'C192' in k})
PY
}
run base; SQD_HIP_LIBRARY=$PWD/scratch/libsqdhip_nt.so run nt; run base2; SQD_HIP_LIBRARY=$PWD/scratch/libsqdhip_nt.so run nt2
SQD_HIP_LIBRARY=$PWD/scratch/libsqdhip_nt.so SQD_PMC_ARGS="--no-pipeline --inflight 1" bash scratch/pmc.sh r05y_nt_fetch FETCH_SIZE > $O/pmc_fetch.log 2>&1; grep -i "conv_wino_kernel<2" gpurun_out/pmc_r05y_nt_fetch.csv
SQD_HIP_LIBRARY=$PWD/scratch/libsqdhip_nt.so SQD_PMC_ARGS="--no-pipeline --inflight 1" bash scratch/pmc.sh r05y_nt_write WRITE_SIZE > $O/pmc_write.log 2>&1; grep -i "conv_wino_kernel<2" gpurun_out/pmc_r05y_nt_write.csv
