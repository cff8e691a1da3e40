#!/bin/bash
TAG=${1:-t}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 $REPO/bench.py --mode train --steps 10 --warmup 3 > $OUT/bench_under_prof.json 2> $OUT/stderr.log
cd $REPO
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/trace_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
nsteps=10+3+3+2+1
print("total GPU kernel time per step ~", tot/1e6/ (10+3+3+2), "ms (approx; steps incl warmup/profile passes)")
for r in rows[:22]: print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  {r["Percentage"]}%')
PY
