#!/bin/bash
# the round's final evidence set in one call: bash scratch/run_final.sh <tag>
TAG=${1:-r05p}
bash scratch/run_profiles.sh $TAG
timeout -k 10 300 python bench.py --arch squeezedetplus --batch 16 --no-cpu-baseline --no-pipeline > gpurun_out/$TAG/bench_squeezedetplus.json 2> gpurun_out/$TAG/bench_squeezedetplus.err; echo "plus rc $?"
