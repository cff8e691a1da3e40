#!/bin/bash
# HBM traffic per kernel: two separate PMC passes (FETCH_SIZE / WRITE_SIZE do not fit one pass), KB units,
# FETCH_SIZE doubled for gfx950 (MI355X_MICROARCH.md, HBM section).  -> gpurun_out/traffic_<tag>.json
TAG=${1:-r01}
export TMPDIR=/tmp
REPO=$PWD
for C in FETCH_SIZE WRITE_SIZE; do
  OUT=$REPO/gpurun_out/pmc_${TAG}_$C; mkdir -p $OUT; cd /tmp
  rocprofv3 --pmc $C --output-format csv -d $OUT -o pmc -- python3 $REPO/bench.py --mode infer --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $OUT/out.json 2> $OUT/stderr.log
  cd $REPO
done
python3 scratch/traffic_aggregate.py $TAG

