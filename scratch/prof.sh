#!/bin/bash
# usage: scratch/prof.sh <tag>   (run on the GPU box from the repo root)
set -e
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 $OLDPWD/bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_prof.json 2> $OUT/stderr.log
cd $OLDPWD
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_$TAG.csv
head -30 gpurun_out/kernel_stats_$TAG.csv
