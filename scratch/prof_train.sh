#!/bin/bash
# usage: scratch/prof_train.sh <tag> [suffix]   (run on the GPU box from the repo root): kernel trace of the training step
# SQD_BENCH_ARGS adds bench.py flags (e.g. "--gpus 1 --force-dist" with suffix _dist: the RCCL gradient exchange in a one-rank group)
set -e
TAG=${1:-r01}
SUF=${2:-}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_train_$TAG$SUF
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 $OLDPWD/bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline $SQD_BENCH_ARGS > $OUT/bench_under_prof.json 2> $OUT/stderr.log
cd $OLDPWD
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_train_$TAG$SUF.csv
head -45 gpurun_out/kernel_stats_train_$TAG$SUF.csv
