#!/bin/bash
# usage (on the GPU box, repo root): bash scratch/run_profiles.sh <tag>  -- the round's evidence in one call:
# full -m gpu log, default bench line, kernel traces of both modes, FETCH/WRITE PMC passes of both modes
TAG=${1:-r03z}
mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/$TAG/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/$TAG/pytest_gpu.log
timeout -k 10 400 python bench.py > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err; echo "bench rc $?"
bash scratch/prof.sh $TAG > gpurun_out/$TAG/prof_infer.log 2>&1; echo "prof rc $?"
bash scratch/prof_train.sh $TAG > gpurun_out/$TAG/prof_train.log 2>&1; echo "prof_train rc $?"
bash scratch/traffic.sh $TAG "infer train" > gpurun_out/$TAG/traffic.log 2>&1; echo "traffic rc $?"
tail -30 gpurun_out/$TAG/traffic.log
