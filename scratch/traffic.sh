#!/bin/bash
# HBM traffic per kernel: two separate PMC passes per mode (FETCH_SIZE / WRITE_SIZE do not fit one pass), KB units,
# FETCH_SIZE doubled for gfx950 (MI355X_MICROARCH.md, HBM section).  usage: scratch/traffic.sh <tag> [modes="infer train"]
#   -> gpurun_out/traffic_<tag>.json  (inference entries at the top level, the training step's under "train"; copy to profiles/traffic.json)
TAG=${1:-r01}
MODES=${2:-"infer train"}
export TMPDIR=/tmp
REPO=$PWD
for MODE in $MODES; do
  for C in FETCH_SIZE WRITE_SIZE; do
    OUT=$REPO/gpurun_out/pmc_${TAG}_${MODE}_$C; mkdir -p $OUT; cd /tmp
    timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT -o pmc -- python3 $REPO/bench.py --mode $MODE --steps 3 --warmup 2 --no-cpu-baseline --no-graph --no-pipeline > $OUT/out.json 2> $OUT/stderr.log || exit 1
    cd $REPO
  done
done
python3 scratch/traffic_aggregate.py $TAG $MODES
