#!/bin/bash
# usage: scratch/prof.sh <tag> [suffix] [extra bench.py args]   (run on the GPU box from the repo root)
# kernel trace of the inference half.  suffix "" = the default command (two lanes in flight: launches of the two lanes overlap, so a
# kernel's average duration mixes contended and uncontended launches); suffix "_serial" with "--inflight 1 --no-pipeline" = one step
# at a time: THIS is the trace whose per-kernel averages reproduce bench.py's HIP-event medians (tests/test_profiles.py).
set -e
TAG=${1:-r01}
SUF=${2:-}
shift; shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG$SUF
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 $OLDPWD/bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/bench_under_prof.json 2> $OUT/stderr.log
cd $OLDPWD
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_$TAG$SUF.csv
head -12 gpurun_out/kernel_stats_$TAG$SUF.csv
