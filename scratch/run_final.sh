#!/bin/bash
# the round's final evidence set in one call: bash scratch/run_final.sh <tag>
TAG=${1:-r05z}
bash scratch/run_profiles.sh $TAG
timeout -k 10 300 python bench.py --arch squeezedetplus --batch 16 --no-cpu-baseline --no-pipeline > gpurun_out/$TAG/bench_squeezedetplus.json 2> gpurun_out/$TAG/bench_squeezedetplus.err; echo "plus rc $?"
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/$TAG/smoke.log 2>&1; echo "smoke rc $?"; tail -1 gpurun_out/$TAG/smoke.log
timeout -k 10 330 python tools/fuzz_conv.py 150 20261006 > gpurun_out/$TAG/fuzz_conv.log 2>&1; echo "fuzz_conv rc $?"; tail -2 gpurun_out/$TAG/fuzz_conv.log
