#!/bin/bash
O=gpurun_out/r05ab; mkdir -p $O
show() { python - $1 $2 <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
L = d['layers']['train']
print(sys.argv[2], 'ms/step', d['ms_per_step'], round(sum(v for k, v in L.items() if 'wgrad' in k and '1tap' in k), 1), 'reduce', [v for k, v in L.items() if 'reduce_batched' in k])
PY
}
for k in 1 2 3; do
timeout -k 10 200 python bench.py --mode train --steps 40 --warmup 10 --no-cpu-baseline --layers > $O/g$k.json 2> $O/g$k.err; show $O/g$k.json grouped
timeout -k 10 200 python scratch/bench_no_e1_group.py --mode train --steps 40 --warmup 10 --no-cpu-baseline --layers > $O/s$k.json 2> $O/s$k.err; show $O/s$k.json single
done
